// ukm_srmerge.hip — merge / union of MANY sorted streams (hundreds to a thousand files) in ONE pass over HBM: the
// MI355X replacement for container/heap's pop/push per k-mer in mergeChunksFile (util-sort.go:196-225, 227-606) and for
// the per-k-mer map probes of an n-file `union` (union.go:186-208) when the files are many and short.
//
// Why: the streaming k-way kernel of ukm_kway.hip merges 8 streams per level, so 1000 files take four levels and every
// record crosses HBM four times (1000 x 1e6 records with taxids: 33 ms for 24 GB of algorithmic traffic).  Its fan-in is
// bound by LDS (a chunk of every child must be resident), not by anything in the data.  With ~1000 streams a value range
// of a few thousand records holds only a handful of records of each stream, so here the roles are swapped:
//
//   * the CODE SPACE is cut into R = N / ~3900 value ranges by splitters from a regular sample of all streams; a range's
//     records (~4 per stream for 1000 streams) fit ONE LDS tile;
//   * `sr_cuts_kernel` finds every stream's slice of every range while it streams each input once, coalesced: a tile of
//     2048 consecutive records of one stream sits in LDS and every splitter that falls into the tile's value span is
//     looked up there (an S x R table of u32 cut points; 1 GB for 1000 files x 1e6 records).  The strict / non-strict
//     order of every stream is checked on the way, so the merge pass does no checking at all;
//   * `sr_merge_kernel`, one workgroup per range: every thread owns two streams (cut points, base pointers and cursors in
//     registers; the output position of the range is the sum of its cut points over the streams — no look-back, no scan),
//     copies their slices into the LDS tile in STREAM ORDER, then orders its VT consecutive tile positions by a stable
//     rank count (branch-free), and
//     log2(threads) rounds of pairwise merge-path merges inside LDS (A before B on ties) finish a stable merge sort of
//     the tile: equal codes stay in stream order, which is the order a stable k-way merge gives.  MERGE writes the tile
//     to its final place; UNION folds every run (TaxId: the left fold of LCAs over a run is its first member when all
//     are equal, 0 when they differ and one has no pre-order number, else the LCA of the members with the smallest and
//     the largest pre-order number: two LDS atomics per record of a run, one table LCA per run) and writes the heads to
//     the range's slot, gathered afterwards by one copy kernel.
//   A range whose slices do not fit the tile (sampling is approximate; one code may be in every file) is worked off in
//   several passes by value: a code v with half a tile to a tile of records below it is found by interpolation between
//   the range's ends (bisection if that does not converge; every probe is one lower bound per stream inside its slice
//   and a block sum), and the pass takes the records below v; a single code with more copies than the tile holds sends
//   the call back to the multi-level merge.
//   Algorithmic bytes: 8 (+4) per input record read + the same per output record written; on top the cut pass reads
//   the keys once more (8 B per record) and the tables cost 8 B per (stream, range).
// Integer, HBM / LDS bound; no MFMA.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "ukm_device.h"
#include "ukm_srmerge.h"

namespace {

constexpr u64 SR_MAX = ~0ull;
enum { SR_FLAG_UNSORTED = 2, SR_FLAG_DEGENERATE = 8 };
enum { SR_NEQ = 1, SR_BAD = 2, SR_DEAD = 4, SR_EXACT = 8 };

constexpr int SR_MAX_STREAMS = 1024;  // two streams per thread, their cursors in registers
constexpr int SR_SAMPLES_PER_RANGE = 128;

// ---- sample: every D-th record of every stream ---------------------------------------------------------------------
__global__ void sr_sample_kernel(const u64 *const *leaf_keys, const u64 *sample_base, u32 S, u64 D, u64 ns, u64 *samples) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ns) return;
    u32 lo = 0, hi = S;  // last j with sample_base[j] <= g
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (sample_base[mid] <= g) lo = mid; else hi = mid;
    }
    const u64 i = g - sample_base[lo];
    // every stream samples at its own phase: files that are subsets of one collection walk through the code space in
    // step, and samples taken at the same indices of every file would arrive in clusters
    const u64 phase = ((u64)lo * 0x9E3779B97F4A7C15ull >> 33) % D;
    samples[g] = leaf_keys[lo][(i + 1) * D - 1 - phase];
}

// how many samples equal their predecessor in the sorted sample: an estimate of how many copies a code has among the streams
__global__ void sr_dupcount_kernel(const u64 *samples, u64 ns, u64 *out) {
    // (grid-stride, one atomic per wave at the end: with one per 64 samples the single counter costs 6 ms at 3e7 samples)
    u32 cnt = 0;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < ns; g += (u64)gridDim.x * blockDim.x)
        cnt += (g > 0 && samples[g] == samples[g - 1]) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += (u32)__shfl_xor((int)cnt, o);
    if (cnt && lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)cnt);
}

// Are the taxids of the streams' records related?  Pairs of records drawn from two random streams at random places: out[0] +=
// pairs, out[1] += pairs with two different taxids of one clade (TaxDev::clade8).  Unrelated taxa: ~ 1 / clades.  Related
// (one species' strains, taxids by clade): most.  Decides SrArgs::clade_emit.
__global__ void sr_taxsample_kernel(const u32 *const *leaf_tax, const u64 *leaf_len, u32 S, TaxDev T, u64 *out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 h = ((u64)i + 1) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const u32 f = (u32)(h % S), g = (u32)((h >> 20) % S);
    const u32 *tf = leaf_tax[f], *tg = leaf_tax[g];
    const u64 lf = leaf_len[f], lg = leaf_len[g];
    bool pair = false, same = false;
    if (tf && tg && lf && lg && f != g) {
        const u64 h2 = h * 0x94D049BB133111EBull;
        const u32 a = tf[(h2 >> 7) % lf], b = tg[(h2 >> 31) % lg];
        pair = true;
        same = a != b && a < T.size && b < T.size && T.clade8[a] != 0 && T.clade8[a] == T.clade8[b];
    }
    const u64 mp = __ballot(pair), ms = __ballot(same);
    if (lane_id() == 0 && mp) {
        atomicAdd((unsigned long long *)&out[0], (unsigned long long)__popcll(mp));
        if (ms) atomicAdd((unsigned long long *)&out[1], (unsigned long long)__popcll(ms));
    }
}

// splitter r (1 <= r < R) = the sample of rank r * ns / R; spl[0] is unused
__global__ void sr_splitters_kernel(const u64 *samples, u64 ns, u32 R, u64 *spl) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    spl[r] = r == 0 ? 0 : samples[r * ns / R];  // (the host keeps R * ns below 2^62)
}

// ---- cut points -----------------------------------------------------------------------------------------------------
// One workgroup per SEGMENT of CT_SEG consecutive tiles of one stream.  Per tile: 2048 records -> LDS (coalesced), order
// check against the predecessor, then every splitter v with key[beg - 1] < v <= key[end - 1] is looked up in the tile:
// lower_bound(stream, v) = beg + lower_bound(tile, v).  Splitters are taken NT at a time from where the previous tile
// stopped (they are sorted), so only a segment's first tile searches the splitter array.
constexpr int CT_NT = 256;
constexpr int CT_VT = 8;
constexpr int CT = CT_NT * CT_VT;
constexpr int CT_SEG = 16;

struct CutArgs {
    const u64 *const *leaf_keys;
    const u64 *leaf_len;
    const u64 *seg_base;  // [S + 1]: segments in front of stream j
    const u64 *spl;       // [R]
    u32 *cuts;            // [S][R + 1]
    u64 *result;          // [1] |= flags
    u32 S, R;
};

__global__ __launch_bounds__(CT_NT) void sr_cuts_kernel(CutArgs p) {
    __shared__ u64 s_k[CT + 2];
    __shared__ u32 s_cnt[CT_NT / 64];
    __shared__ u32 s_r;
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    // stream of this segment (uniform: scalar loads)
    u32 lo = 0, hi = p.S;
    const u64 g = blockIdx.x;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (sload_u64(p.seg_base + mid) <= g) lo = mid; else hi = mid;
    }
    const u32 j = lo;
    const u64 seg = g - sload_u64(p.seg_base + j);
    const u64 n = sload_u64(p.leaf_len + j);
    const u64 *k = p.leaf_keys[j];
    const u32 R = p.R;
    u32 *crow = p.cuts + (size_t)j * ((size_t)R + 1);
    const u64 ntiles = (n + CT - 1) / CT;
    const u64 t0 = seg * CT_SEG, t1 = (t0 + CT_SEG < ntiles) ? t0 + CT_SEG : ntiles;
    if (tid == 0) {
        if (seg == 0) crow[0] = 0;
        if (t1 == ntiles) crow[R] = (u32)n;
        // first splitter of this segment: the first r >= 1 with spl[r] > key[beg - 1]
        u32 ra = 1;
        if (t0 > 0 && R > 1) {
            const u64 pk = k[t0 * CT - 1];
            u32 a = 1, b = R;  // first r in [1, R) with spl[r] > pk, else R
            while (a < b) {
                const u32 mid = (a + b) >> 1;
                if (p.spl[mid] > pk) b = mid; else a = mid + 1;
            }
            ra = a;
        }
        s_r = ra;
    }
    __syncthreads();
    u32 rc = s_r;  // next splitter to place (the same in every thread)
    u32 bad = 0;
    for (u64 t = t0; t < t1; t++) {
        const u64 beg = t * CT;
        const int cnt = (int)((n - beg < (u64)CT) ? (n - beg) : (u64)CT);
        __syncthreads();  // the previous tile's searches are done
#pragma unroll
        for (int s = 0; s < CT_VT; s++) {
            const int x = tid + s * CT_NT;
            if (x < cnt) s_k[x + 1] = k[beg + x];
        }
        if (tid == 0) s_k[0] = beg > 0 ? k[beg - 1] : 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < CT_VT; s++) {
            const int x = tid + s * CT_NT;
            if (x < cnt && (x > 0 || beg > 0) && s_k[x] > s_k[x + 1]) bad = SR_FLAG_UNSORTED;
        }
        const bool last = t + 1 == ntiles;
        const u64 lastkey = s_k[cnt];
        for (;;) {
            const u32 r = rc + (u32)tid;
            bool mine = false;
            if (r < R) {
                const u64 v = p.spl[r];
                mine = last || v <= lastkey;
                if (mine) {
                    int a = 0, b = cnt;  // first x with tile[x] >= v
                    while (a < b) {
                        const int mid = (a + b) >> 1;
                        if (s_k[mid + 1] < v) a = mid + 1; else b = mid;
                    }
                    crow[r] = (u32)(beg + (u64)a);
                }
            }
            const u64 m = __ballot(mine);
            if (lane == 0) s_cnt[wave] = (u32)__popcll(m);
            __syncthreads();
            u32 tot = 0;
#pragma unroll
            for (int w = 0; w < CT_NT / 64; w++) tot += s_cnt[w];
            __syncthreads();
            rc += tot;
            if (tot < (u32)CT_NT) break;
        }
    }
    if (__ballot(bad != 0) && lane == 0) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)SR_FLAG_UNSORTED);
}

// ---- the merge ---------------------------------------------------------------------------------------------------------
struct SrArgs {
    const u64 *const *leaf_keys;  // [S]
    const u32 *const *leaf_tax;   // [S] (entries may be null) or nullptr
    const u32 *cuts;              // [S][R + 1]
    const u64 *spl;               // [R]: range r holds the codes in [spl[r], spl[r + 1]) (r = 0: from 0; r = R - 1: to the end)
    u64 *out_keys;                // MERGE: the caller's buffer; UNION: slots of N records
    u32 *out_tax;
    u64 *out_cnt;                 // [R] (UNION)
    u64 *out_off;                 // [R] (UNION): where the range's slot begins
    u64 *result;                  // [1] |= flags
    u32 S, R, per_xcd;
    u32 threshold;                // UNION: > 1 = only codes with at least this many records (`common`)
    u32 buckets;                  // 1: tiles are ordered by counting placement (files that share next to nothing), 0: by the merge rounds
    u32 clade_emit;               // UNION with taxids: the runs' taxids are folded by their one-byte clade codes first (round 5; see the emit)
    TaxDev tax;
};

// One round of the in-LDS merge sort: thread `lt` of a pair merges VT consecutive positions of merge(A, B) (A first on
// ties) into registers.  Both runs are followed by an SR_MAX sentinel; positions >= la + lb receive garbage nobody
// reads.  With TaxIds the thread records the LDS slot every output came from and fetches the nine taxids afterwards
// (independent reads), instead of one more dependent read and two more selects per step.
// The rounds are bound by VALU issue (PMC: 27 lane-instructions per record and round), so the step is the 12-instruction
// one of ukm_kway.hip: ONE compare feeds the minimum, the cursor and the head update, only the B cursor is tracked (in
// bytes).  (Measured and dropped: two heads per side in registers so that a step's LDS read is not needed before the step
// after next -- 17 instructions per step and no faster: LDS latency is not what the rounds wait for.)
template <int VT>
__device__ __forceinline__ int sr_merge_path(const u64 *in, int abase, int la, int bbase, int lb, int diag) {
    // byte offsets: two adds per probe instead of shifts and index arithmetic
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
    return lo8 >> 3;
}

template <bool TAX, int VT>
__device__ __forceinline__ void sr_merge_fast(const u64 *in, const u32 *tin, int abase, int la, int bbase, int lb, int lt,
                                              u64 (&ro)[VT], u32 (&rt)[VT]) {
    const int L = la + lb;
    int diag = lt * VT;
    diag = diag < L ? diag : L;
    const int lo8 = sr_merge_path<VT>(in, abase, la, bbase, lb, diag) * 8;
    const char *inb = reinterpret_cast<const char *>(in);
    const int sum8 = (abase + bbase + diag) * 8;  // A cursor + B cursor, in bytes, grows by one record per step
    int pb8 = (bbase + diag) * 8 - lo8;
    u64 ak = *reinterpret_cast<const u64 *>(inb + abase * 8 + lo8), bk = *reinterpret_cast<const u64 *>(inb + pb8);
    int src8[VT];
#pragma unroll
    for (int s = 0; s < VT; s++) {
        const bool take_b = bk < ak;
        ro[s] = take_b ? bk : ak;
        asm volatile("" : "+v"(ro[s]));
        if (TAX) src8[s] = take_b ? pb8 : sum8 + 8 * s - pb8;
        pb8 += take_b ? 8 : 0;
        const int idx8 = take_b ? pb8 : sum8 + 8 * (s + 1) - pb8;
        const u64 nk = *reinterpret_cast<const u64 *>(inb + idx8);
        ak = take_b ? ak : nk;
        bk = take_b ? nk : bk;
    }
    if (TAX) {
        const char *tb = reinterpret_cast<const char *>(tin);
#pragma unroll
        for (int s = 0; s < VT; s++) rt[s] = *reinterpret_cast<const u32 *>(tb + (src8[s] >> 1));
    }
}

// the same with exhaustion tested by index: for a tile that holds a real 2^64-1 code (the sentinel's value)
template <bool TAX, int VT>
__device__ __forceinline__ void sr_merge_checked(const u64 *in, const u32 *tin, int abase, int la, int bbase, int lb, int lt,
                                                 u64 (&ro)[VT], u32 (&rt)[VT]) {
    const int L = la + lb;
    int diag = lt * VT;
    diag = diag < L ? diag : L;
    const int lo = sr_merge_path<VT>(in, abase, la, bbase, lb, diag);
    int pa = abase + lo, pb = bbase + diag - lo;
    const int ea = abase + la, eb = bbase + lb;
    u64 ak = in[pa], bk = in[pb];
#pragma unroll
    for (int s = 0; s < VT; s++) {
        const bool take_a = (pa < ea) && (pb >= eb || ak <= bk);
        ro[s] = take_a ? ak : bk;
        if (TAX) rt[s] = tin[take_a ? pa : pb];
        pa += take_a ? 1 : 0;
        pb += take_a ? 0 : 1;
        const u64 nk = in[take_a ? pa : pb];
        ak = take_a ? nk : ak;
        bk = take_a ? bk : nk;
    }
}

template <int NT>
__device__ __forceinline__ u32 sr_block_sum_u32(u32 v, u32 *smem) {
    u32 tot;
    (void)block_excl_scan_u32<NT>(v, smem, &tot);
    return tot;
}

template <int NT>
__device__ __forceinline__ u64 sr_block_min_u64(u64 v, u64 *smem) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    if (lane_id() == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 m = smem[0];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) m = smem[w] < m ? smem[w] : m;
    __syncthreads();
    return m;
}

template <int NT>
__device__ __forceinline__ u64 sr_block_sum_u64(u64 v, u64 *smem) {
    v = wave_reduce_sum_u64(v);
    if (lane_id() == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 m = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) m += smem[w];
    __syncthreads();
    return m;
}

#ifndef SR_WAVES_TAX
#define SR_WAVES_TAX 4
#endif
#ifndef SR_WAVES_PLAIN
#define SR_WAVES_PLAIN 4
#endif

#ifdef SR_PHASES  // developer build: thread 0 of every workgroup adds the cycles of each phase to result[8 + k]
#define SR_PH(k)                                                                                  \
    do {                                                                                          \
        if (threadIdx.x == 0) {                                                                   \
            const long long _c = clock64();                                                       \
            atomicAdd((unsigned long long *)&p.result[8 + (k)], (unsigned long long)(_c - ph_t)); \
            ph_t = _c;                                                                            \
        }                                                                                         \
    } while (0)
#else
#define SR_PH(k) do { } while (0)
#endif

// first record >= v (UPPER = false) / > v (UPPER = true) of k[a, b)
template <bool UPPER>
__device__ __forceinline__ u32 sr_bound(const u64 *k, u32 a, u32 b, u64 v) {
    while (a < b) {
        const u32 mid = a + ((b - a) >> 1);
        const u64 x = *as_global(k + mid);
        const bool below = UPPER ? x <= v : x < v;
        a = below ? mid + 1 : a;
        b = below ? b : mid;
    }
    return a;
}

typedef u64 sr_u64x2 __attribute__((ext_vector_type(2), aligned(8)));
typedef u32 sr_u32x4 __attribute__((ext_vector_type(4), aligned(4)));

#ifndef SR_SPT_N
#define SR_SPT_N 2
#endif
#ifndef SR_NB
#define SR_NB 4096          /* value buckets of the counting placement */
#endif
#ifndef SR_BUCKET_LIMIT
#define SR_BUCKET_LIMIT 15  /* a fuller bucket: the tile takes the merge rounds (the arrival number has four bits) */
#endif
constexpr int SR_SPT = SR_SPT_N;  // streams per thread (their cursors and pointers live in registers): S <= NT * SR_SPT


// ---- counting placement by dense code numbers (round 5) ------------------------------------------------------------------
// Files that share their codes put few DISTINCT codes into a tile of NT * VT records (100 copies of every code: ~46; five
// copies: ~900).  The records go to registers and the tile's LDS becomes scratch: a hash table collects the distinct codes,
// these are ranked (counted into 512 value buckets, staged bucket by bucket, a code's rank = its bucket's place + the
// smaller codes of its bucket: the counting placement of ORD 1, over a few hundred codes instead of all records), and a
// record's final place is [records of smaller codes] + [records of its code in earlier waves] + [... in earlier rows of
// its wave] + [... on lower lanes of its row] -- the order a stable merge gives, because a wave's rows are consecutive
// tile positions.  The last term comes from ten ballots (a rank has ten bits), not from a loop over the row's distinct
// codes.  ~130 lane-instructions per record and 12 barriers instead of the ~280 and 20 of the rank sort + nine merge
// rounds.  A tile with more than SR_DMAX distinct codes (or one whose table probes run long, or that holds the code
// 2^64 - 1, the table's empty mark) is put back and takes the merge rounds.
constexpr int SR_HT = 2048;    // table slots
constexpr int SR_DMAX = 1024;  // distinct codes a tile may hold (a rank has ten bits)
constexpr int SR_DBITS = 10;
constexpr int SR_DNB = 512;    // value buckets of the ranking
constexpr int SR_DENSE_WORDS = SR_HT / 2 + SR_DMAX / 2 + 2 * SR_DNB + 1;  // 32-bit words of the second scratch region

template <bool TAX, int NT, int VT>
__device__ __forceinline__ bool sr_place_dense(u64 *s_key, u32 *s_tax, u32 *s_b, u32 *s_scan, u64 *s_red, u32 n, bool danger) {
    constexpr int NW = NT / 64;
    static_assert(SR_HT + SR_DMAX + NW * SR_DMAX / 4 <= NT * VT + NT + VT + 8, "scratch");
    static_assert(SR_DNB == NT && SR_DMAX == 2 * NT, "one bucket, two ranks per thread");
    if (n <= (u32)VT) return false;  // (uniform)
    u64 *ht = s_key;                                                         // [SR_HT] the distinct codes
    u64 *stg = ht + SR_HT;                                                   // [SR_DMAX] ... staged bucket by bucket
    unsigned short *wc = reinterpret_cast<unsigned short *>(stg + SR_DMAX); // [NW][SR_DMAX] records per (wave, rank)
    unsigned short *sid = reinterpret_cast<unsigned short *>(s_b);          // [SR_HT] slot -> rank of its code
    unsigned short *sslot = sid + SR_HT;                                     // [SR_DMAX] staged entry -> slot
    u32 *bcnt = s_b + SR_HT / 2 + SR_DMAX / 2;                               // [SR_DNB] codes per bucket, then the staging cursor
    u32 *bbase = bcnt + SR_DNB;                                              // [SR_DNB + 1] the buckets' places
    u32 *ctl = reinterpret_cast<u32 *>(s_red);                               // [0] distinct codes [1] gave up; s_red[2] min, [3] max
    int tr = (int)threadIdx.x;
    asm volatile("" : "+v"(tr));
    const int w = tr >> 6, lane = tr & 63;
    const int q0 = w * (VT * 64) + lane;  // this thread's records: tile positions q0 + 64 t
    u64 kk[VT];
    u32 tt[VT];
    u64 mn = SR_MAX, mx = 0;
#pragma unroll
    for (int t = 0; t < VT; t++) {
        const bool valid = q0 + 64 * t < (int)n;
        kk[t] = valid ? s_key[q0 + 64 * t] : SR_MAX;
        if (TAX) tt[t] = valid ? s_tax[q0 + 64 * t] : 0u;
        mn = kk[t] < mn ? kk[t] : mn;
        mx = (valid && kk[t] > mx) ? kk[t] : mx;
    }
    if (tr < 2) s_red[tr] = 0;
    if (tr == 2) s_red[2] = SR_MAX;
    if (tr == 3) s_red[3] = 0;
    if (__syncthreads_or(danger ? 1 : 0)) return false;  // (nothing has been overwritten yet)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o1 = __shfl_xor(mn, d, 64), o2 = __shfl_xor(mx, d, 64);
        mn = o1 < mn ? o1 : mn;
        mx = o2 > mx ? o2 : mx;
    }
    if (lane == 0) {
        atomicMin(reinterpret_cast<unsigned long long *>(&s_red[2]), (unsigned long long)mn);
        atomicMax(reinterpret_cast<unsigned long long *>(&s_red[3]), (unsigned long long)mx);
    }
    for (int i = tr; i < SR_HT; i += NT) ht[i] = SR_MAX;
    for (int i = tr; i < NW * SR_DMAX / 2; i += NT) reinterpret_cast<u32 *>(wc)[i] = 0;
    bcnt[tr] = 0;
    __syncthreads();
    u32 sl[VT];
    {
        u32 fresh = 0;
        bool gave_up = false;
#pragma unroll
        for (int t = 0; t < VT; t++) {
            sl[t] = 0;
            if (q0 + 64 * t < (int)n) {
                const u32 lo32 = (u32)kk[t], hi32 = (u32)(kk[t] >> 32);
                u32 h = ((lo32 ^ ((hi32 << 15) | (hi32 >> 17)) ^ (hi32 >> 3)) * 0x9E3779B1u) >> 21;
                int probe = 0;
                for (; probe < 48; probe++) {
                    const u64 prev = atomicCAS(reinterpret_cast<unsigned long long *>(&ht[h]), (unsigned long long)SR_MAX, (unsigned long long)kk[t]);
                    if (prev == SR_MAX) {
                        fresh++;
                        break;
                    }
                    if (prev == kk[t]) break;
                    h = (h + 1) & (u32)(SR_HT - 1);
                }
                gave_up = gave_up || probe == 48;
                sl[t] = h;
            }
        }
        if (fresh) atomicAdd(&ctl[0], fresh);
        if (gave_up) ctl[1] = 1;
    }
    __syncthreads();
    const u32 D = ctl[0];
    if (D > (u32)SR_DMAX || ctl[1]) {  // (uniform) put the tile back: the merge rounds take it
#pragma unroll
        for (int t = 0; t < VT; t++)
            if (q0 + 64 * t < (int)n) {
                s_key[q0 + 64 * t] = kk[t];
                if (TAX) s_tax[q0 + 64 * t] = tt[t];  // (the second scratch region)
            }
        __syncthreads();
        return false;
    }
    // ---- ranks of the distinct codes ------------------------------------------------------------------------------------
    mn = s_red[2];
    mx = s_red[3];
    const float sc = (float)SR_DNB / ((float)(mx - mn) + 1.0f);  // bucket = floor((code - mn) * SR_DNB / span): monotone in the code
    auto bucket_of = [&](u64 k) -> u32 {
        const u32 b = (u32)((float)(k - mn) * sc);
        return b < (u32)SR_DNB - 1u ? b : (u32)SR_DNB - 1u;
    };
    for (int i = tr; i < SR_HT; i += NT) {
        const u64 k = ht[i];
        if (k != SR_MAX) atomicAdd(&bcnt[bucket_of(k)], 1u);
    }
    __syncthreads();
    {
        u32 all;
        const u32 mine = bcnt[tr];
        const u32 ex = block_excl_scan_u32<NT>(mine, s_scan, &all);
        bbase[tr] = ex;
        bcnt[tr] = 0;
        if (tr == NT - 1) bbase[SR_DNB] = D;
    }
    __syncthreads();
    for (int i = tr; i < SR_HT; i += NT) {
        const u64 k = ht[i];
        if (k != SR_MAX) {
            const u32 b = bucket_of(k);
            const u32 at = bbase[b] + atomicAdd(&bcnt[b], 1u);
            stg[at] = k;
            sslot[at] = (unsigned short)i;
        }
    }
    __syncthreads();
    for (int e = tr; e < (int)D; e += NT) {
        const u64 k = stg[e];
        const u32 b = bucket_of(k);
        const u32 beg = bbase[b], end = bbase[b + 1];
        u32 smaller = 0;
        for (u32 j = beg; j < end; j++) smaller += stg[j] < k ? 1u : 0u;
        sid[sslot[e]] = (unsigned short)(beg + smaller);
    }
    __syncthreads();
    // ---- rows of a wave, one after the other: place inside (wave, code) = records of the code in earlier rows + on lower lanes
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int t = 0; t < VT; t++) {
        const bool valid = q0 + 64 * t < (int)n;
        const u32 id = valid ? (u32)sid[sl[t]] : 0u;
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < SR_DBITS; b++) {
            const bool bit = (id >> b) & 1u;
            const u64 B = __ballot(bit);
            m &= bit ? B : ~B;
        }
        u32 place = 0;
        if (valid) {
            const u32 before = (u32)__popcll(m & lt), all = (u32)__popcll(m);
            unsigned short *cp = &wc[w * SR_DMAX + (int)id];
            const u32 base = *cp;
            if (before == 0) *cp = (unsigned short)(base + all);  // (LDS operations of a wave complete in order)
            place = base + before;
        }
        sl[t] = id | (place << 16);
    }
    __syncthreads();
    {
        // ranks 2 tr and 2 tr + 1: records per wave -> places of the (rank, wave) groups
        u32 tot0 = 0, tot1 = 0;
        unsigned short *c0 = &wc[2 * tr];
#pragma unroll
        for (int v = 0; v < NW; v++) {
            const u32 pair = *reinterpret_cast<u32 *>(c0 + v * SR_DMAX);
            *reinterpret_cast<u32 *>(c0 + v * SR_DMAX) = tot0 | (tot1 << 16);
            tot0 += pair & 0xFFFFu;
            tot1 += pair >> 16;
        }
        u32 all;
        const u32 ex = block_excl_scan_u32<NT>(tot0 + tot1, s_scan, &all);
        const u32 add = ex | ((ex + tot0) << 16);
#pragma unroll
        for (int v = 0; v < NW; v++) *reinterpret_cast<u32 *>(c0 + v * SR_DMAX) += add;  // (both halves stay below 2^16: no carry)
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < VT; t++) sl[t] = (u32)wc[w * SR_DMAX + (int)(sl[t] & 0xFFFFu)] + (sl[t] >> 16);
    __syncthreads();  // the scratch has been read: the tile is written in its final order
#pragma unroll
    for (int t = 0; t < VT; t++)
        if (q0 + 64 * t < (int)n) {
            s_key[sl[t]] = kk[t];
            if (TAX) s_tax[sl[t]] = tt[t];
        }
    __syncthreads();
    return true;
}

// ORD: how a tile is brought into stable order -- 0 the merge rounds, 1 counting placement by value buckets (files that
// share next to nothing) with the rounds behind it, 2 counting placement by the DENSE NUMBER of a record's code among
// the tile's distinct codes (files that share a lot: a tile holds a few hundred distinct codes) with the rounds behind it
template <bool TAX, bool UNION, int NT, int VT, int LOGNT, int ORD>
__global__ __launch_bounds__(NT, (TAX ? SR_WAVES_TAX : SR_WAVES_PLAIN)) void sr_merge_kernel(SrArgs p) {
#ifdef SR_PHASES
    long long ph_t = clock64();
#endif
    constexpr int CAP = NT * VT;
    constexpr int BUF = CAP + NT + VT + 8;
    static_assert((1 << LOGNT) == NT, "NT");
    __shared__ __attribute__((aligned(16))) u64 s_key[BUF];
    __shared__ __attribute__((aligned(16))) u32 s_tax[TAX ? BUF : 4];
    __shared__ u32 s_scan[NT / 64 + 1];
    __shared__ u64 s_red[NT / 64];
    __shared__ u32 s_bkt[ORD == 1 ? SR_NB / 2 + 1 : 1];   // counting placement: records per value bucket (two 16-bit counters per word), then the buckets' places
    __shared__ unsigned short s_ba[ORD == 1 ? BUF : 2];   // ... a record's bucket << 4 | arrival number, later its final place
    __shared__ u32 s_aux[(ORD == 1 && !TAX) ? BUF : ((ORD == 2 && !TAX) ? SR_DENSE_WORDS : 4)];  // ... the staged records' tile positions (with taxids: s_tax, free at that point); ORD 2: its second scratch region

    const int tid = (int)threadIdx.x;
    const u32 r = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);  // an XCD works through CONSECUTIVE ranges: the
    if (r >= p.R) return;                                              // slices of neighbouring ranges share lines in its L2
    const u32 S = p.S;
    const size_t RP = (size_t)p.R + 1;
    // ---- this thread's streams: slice of the range, base pointers ----------------------------------------------------------
    u32 lo[SR_SPT], hi[SR_SPT];
    const u64 *kp[SR_SPT];
    const u32 *tp[SR_SPT];
#pragma unroll
    for (int q = 0; q < SR_SPT; q++) {
        const u32 j = (u32)tid * SR_SPT + (u32)q;
        lo[q] = hi[q] = 0;
        kp[q] = nullptr;
        tp[q] = nullptr;
        if (j < S) {
            const u32 c0 = p.cuts[(size_t)j * RP + r], c1 = p.cuts[(size_t)j * RP + r + 1];
            lo[q] = c0;
            hi[q] = c1 > c0 ? c1 : c0;
            kp[q] = p.leaf_keys[j];
            if (TAX && p.leaf_tax) tp[q] = p.leaf_tax[j];
        }
    }
    // records below the range = where its output begins
    u64 out_pos;
    {
        u64 below = 0;
#pragma unroll
        for (int q = 0; q < SR_SPT; q++) below += lo[q];
        out_pos = sr_block_sum_u64<NT>(below, s_red);
    }
    const u64 out_pos0 = out_pos;
    u64 va = r > 0 ? sload_u64(p.spl + r) : 0;  // no record of what is left lies below va

    for (;;) {
        // ---- what is left of the range; a pass by value if it does not fit the tile -------------------------------------
        u32 e[SR_SPT];
        u32 loc = 0;
#pragma unroll
        for (int q = 0; q < SR_SPT; q++) {
            e[q] = hi[q];
            loc += hi[q] - lo[q];
        }
        u32 n;
        u32 excl = block_excl_scan_u32<NT>(loc, s_scan, &n);
        if (n == 0) break;
#ifdef SR_PHASES
        if (tid == 0 && out_pos == out_pos0) {
            atomicMax((unsigned long long *)&p.result[16], (unsigned long long)n);
            if (n > (u32)CAP) atomicAdd((unsigned long long *)&p.result[17], 1ull);
            if (n > (u32)CAP * 2) atomicAdd((unsigned long long *)&p.result[18], 1ull);
        }
#endif
        bool more = false;
        if (n > (u32)CAP) {
            more = true;
#ifdef SR_PHASES
            if (tid == 0) atomicAdd((unsigned long long *)&p.result[14], 1ull);
#endif
            // A value v with CAP / 2 <= #records below v <= CAP: interpolation between the bracket's ends (codes are
            // spread evenly inside a range, or the range would not be this small), bisection when that does not
            // converge.  Bracket: ca = #records below a (0 at the start), cb = #records below b (> CAP).
            u64 a = va, b = (r + 1 < p.R) ? sload_u64(p.spl + r + 1) : SR_MAX;
            u32 ca = 0, cb = n;
            u32 ea[SR_SPT], eb[SR_SPT];  // per stream: first record >= a, first record >= b
#pragma unroll
            for (int q = 0; q < SR_SPT; q++) {
                ea[q] = lo[q];
                eb[q] = hi[q];
            }
            if (b == SR_MAX || b <= a) {  // the last range (open end) or nothing known: the largest code that is left
                u64 mx = 0;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++)
                    if (hi[q] > lo[q]) {
                        const u64 x = *as_global(kp[q] + (hi[q] - 1));
                        mx = x > mx ? x : mx;
                    }
                mx = ~sr_block_min_u64<NT>(~mx, s_red);
                b = mx;  // #records below mx <= n; if that fits, the pass takes them and the copies of mx follow
                u32 cl = 0;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++) {
                    eb[q] = hi[q] > lo[q] ? sr_bound<false>(kp[q], lo[q], hi[q], b) : lo[q];
                    cl += eb[q] - lo[q];
                }
                cb = sr_block_sum_u32<NT>(cl, s_scan);
            }
            int it = 0;
            while (cb > (u32)CAP) {
                if (b - a <= 1) break;  // every record below b equals a
                u64 v;
                const u64 span = b - a;
                if (it < 4) {
                    const double f = ((double)(CAP - CAP / 8) - (double)ca) / ((double)cb - (double)ca);
                    v = a + (u64)((double)span * (f < 0.0 ? 0.0 : f));
                } else {
                    v = a + (span >> 1);
                }
                v = v <= a ? a + 1 : (v >= b ? b - 1 : v);
                it++;
                u32 ev[SR_SPT];
                u32 cl = 0;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++) {
                    ev[q] = eb[q] > ea[q] ? sr_bound<false>(kp[q], ea[q], eb[q], v) : ea[q];
                    cl += ev[q] - lo[q];
                }
                const u32 c = sr_block_sum_u32<NT>(cl, s_scan);
                if (c > (u32)CAP) {
                    b = v;
                    cb = c;
#pragma unroll
                    for (int q = 0; q < SR_SPT; q++) eb[q] = ev[q];
                } else {
                    a = v;
                    ca = c;
#pragma unroll
                    for (int q = 0; q < SR_SPT; q++) ea[q] = ev[q];
                    if (c >= (u32)CAP / 2) break;
                }
            }
            if (cb <= (u32)CAP) {  // (the open-ended case: everything below the largest code fits)
                a = b;
                ca = cb;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++) ea[q] = eb[q];
            }
            if (ca == 0) {
                // no record below a + 1 ... and more than a tile below b: the smallest code that is left has many copies.
                // All of them, from all streams, if they fit; else the multi-level merge answers (it streams such runs)
                u64 mn = SR_MAX;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++)
                    if (hi[q] > lo[q]) {
                        const u64 x = *as_global(kp[q] + lo[q]);
                        mn = x < mn ? x : mn;
                    }
                mn = sr_block_min_u64<NT>(mn, s_red);
                u32 cl = 0;
#pragma unroll
                for (int q = 0; q < SR_SPT; q++) {
                    ea[q] = hi[q] > lo[q] ? sr_bound<true>(kp[q], lo[q], hi[q], mn) : lo[q];
                    cl += ea[q] - lo[q];
                }
                ca = sr_block_sum_u32<NT>(cl, s_scan);
                if (ca > (u32)CAP) {
                    if (tid == 0) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)SR_FLAG_DEGENERATE);
                    return;
                }
                a = mn;  // (the next pass starts above mn: nothing below it is left)
            }
            va = a;
            loc = 0;
#pragma unroll
            for (int q = 0; q < SR_SPT; q++) {
                e[q] = ea[q];
                loc += e[q] - lo[q];
            }
            excl = block_excl_scan_u32<NT>(loc, s_scan, &n);
        }
        SR_PH(0);

        // ---- gather: a lane copies its streams' slices (a few records each) to the tile, in stream order ------------------
        bool danger = false;
        {
            u32 base = excl;
#pragma unroll
            for (int q = 0; q < SR_SPT; q++) {
                const u32 len = e[q] - lo[q];
                const u64 *src = kp[q] + lo[q];
                // a slice is a few consecutive records: 16-byte loads (two codes / four taxids each; the addresses are
                // only 8- / 4-byte aligned), four codes in flight -- one cache-line look-up per load instead of one per
                // record is what the gather is bound by
                u32 x = 0;
                for (; x + 4 <= len; x += 4) {
                    const sr_u64x2 v0 = *reinterpret_cast<ukm_gptr<sr_u64x2>>((uintptr_t)(src + x));
                    const sr_u64x2 v1 = *reinterpret_cast<ukm_gptr<sr_u64x2>>((uintptr_t)(src + x + 2));
                    s_key[base + x] = v0.x; s_key[base + x + 1] = v0.y; s_key[base + x + 2] = v1.x; s_key[base + x + 3] = v1.y;
                    danger = danger || v1.y == SR_MAX;
                }
                if (x + 2 <= len) {
                    const sr_u64x2 v0 = *reinterpret_cast<ukm_gptr<sr_u64x2>>((uintptr_t)(src + x));
                    s_key[base + x] = v0.x; s_key[base + x + 1] = v0.y;
                    danger = danger || v0.y == SR_MAX;
                    x += 2;
                }
                if (x < len) {
                    const u64 k = *as_global(src + x);
                    s_key[base + x] = k;
                    danger = danger || k == SR_MAX;
                }
                if (TAX) {
                    if (tp[q]) {
                        const u32 *ts = tp[q] + lo[q];
                        u32 y = 0;
                        for (; y + 4 <= len; y += 4) {
                            const sr_u32x4 v = *reinterpret_cast<ukm_gptr<sr_u32x4>>((uintptr_t)(ts + y));
                            s_tax[base + y] = v.x; s_tax[base + y + 1] = v.y; s_tax[base + y + 2] = v.z; s_tax[base + y + 3] = v.w;
                        }
                        for (; y < len; y++) s_tax[base + y] = *as_global(ts + y);
                    } else {
                        for (u32 y = 0; y < len; y++) s_tax[base + y] = 0u;
                    }
                }
                base += len;
            }
        }
        __syncthreads();
        SR_PH(1);
        // ---- the tile in stable order, the cheap way (round 5): COUNTING PLACEMENT by value buckets -------------------------
        // A tile of files that share little holds ~n distinct codes spread evenly over its value span, so SR_NB = 4096
        // buckets of equal width take about one record each.  Every record counts itself into its bucket (one LDS atomic),
        // a scan of the counters gives the buckets' places, the records are staged bucket by bucket, and a record's final
        // place is its bucket's place + the number of records of the same bucket that come before it by (code, tile
        // position) -- the order a stable merge gives -- found by walking the bucket (1 - 2 records).  ~60 lane-instructions
        // per record instead of the ~280 of the per-thread rank sort + nine merge rounds below, and 6 barriers instead of 20.
        // Codes that many files share crowd one bucket (the walk is quadratic in its size): a tile whose fullest bucket
        // holds more than SR_BUCKET_LIMIT records takes the merge rounds, which do not care.
        bool placed = false;
        if constexpr (ORD == 2) placed = sr_place_dense<TAX, NT, VT>(s_key, s_tax, TAX ? s_tax : s_aux, s_scan, s_red, n, danger);
        if constexpr (ORD == 1)
        if (n > (u32)VT) {  // (ORD: the host's choice, from the share of equal neighbours in its sorted sample)
            // (what a record needs between the phases -- its bucket, its arrival number, later its final place -- lives in
            //  the 16-bit array s_ba, and the counters are 16-bit halves of 32-bit words: the records themselves are the
            //  only thing a thread keeps in registers across the barriers, as in the merge rounds)
            auto bkt_get = [&](u32 b) -> u32 {
                const u32 w = s_bkt[b >> 1];
                return (b & 1u) ? (w >> 16) : (w & 0xFFFFu);
            };
            int tr = tid;
            asm volatile("" : "+v"(tr));
            const int p0 = tr * VT;
            u64 mn = SR_MAX, mx = 0;
#pragma unroll
            for (int i = 0; i < VT; i++) {
                const bool valid = p0 + i < (int)n;
                const u64 k = valid ? s_key[p0 + i] : SR_MAX;
                mn = k < mn ? k : mn;                            // (an invalid slot counts as 2^64 - 1: no effect on the minimum)
                mx = (valid && k > mx) ? k : mx;
            }
            for (int i = tid; i <= SR_NB / 2; i += NT) s_bkt[i] = 0;
            mn = sr_block_min_u64<NT>(mn, s_red);
            mx = ~sr_block_min_u64<NT>(~mx, s_red);
            // bucket = floor((code - mn) * SR_NB / (mx - mn + 1)) in single precision: monotone in the code (conversion,
            // product with a positive constant and truncation all are), which is all the order needs
            const float sc = (float)SR_NB / ((float)(mx - mn) + 1.0f);
            u32 worst = 0;
#pragma unroll
            for (int i = 0; i < VT; i++) {
                if (p0 + i < (int)n) {
                    u32 b = (u32)((float)(s_key[p0 + i] - mn) * sc);
                    b = b < (u32)SR_NB - 1u ? b : (u32)SR_NB - 1u;
                    const u32 w = atomicAdd(&s_bkt[b >> 1], (b & 1u) ? 0x10000u : 1u);
                    const u32 arr = (b & 1u) ? (w >> 16) : (w & 0xFFFFu);  // its arrival number in the bucket
                    worst = arr > worst ? arr : worst;
                    s_ba[p0 + i] = (unsigned short)((arr < 15u ? arr : 15u) | (b << 4));
                }
            }
            worst = (u32)__syncthreads_or(worst >= (u32)SR_BUCKET_LIMIT ? 1 : 0);
            if (!worst) {
                // the buckets' places: exclusive scan of the counters (SR_NB / NT consecutive counters per thread)
                constexpr int WPT = SR_NB / 2 / NT;  // counter words per thread
                u32 cw[WPT], sum = 0;
#pragma unroll
                for (int q = 0; q < WPT; q++) {
                    cw[q] = s_bkt[tid * WPT + q];
                    sum += (cw[q] & 0xFFFFu) + (cw[q] >> 16);
                }
                u32 tot;
                u32 ex = block_excl_scan_u32<NT>(sum, s_scan, &tot);
#pragma unroll
                for (int q = 0; q < WPT; q++) {
                    const u32 c0 = cw[q] & 0xFFFFu, c1 = cw[q] >> 16;
                    s_bkt[tid * WPT + q] = ex | ((ex + c0) << 16);
                    ex += c0 + c1;
                }
                if (tid == NT - 1) s_bkt[SR_NB / 2] = n;
                u64 kk[VT];
                u32 tt[VT];
#pragma unroll
                for (int i = 0; i < VT; i++) {
                    const bool valid = p0 + i < (int)n;
                    kk[i] = valid ? s_key[p0 + i] : SR_MAX;
                    if (TAX) tt[i] = valid ? s_tax[p0 + i] : 0u;
                }
                __syncthreads();
                // staged bucket by bucket in arrival order: the code and the record's tile position (the tie-break)
                u32 *s_pos = TAX ? s_tax : s_aux;
#pragma unroll
                for (int i = 0; i < VT; i++)
                    if (p0 + i < (int)n) {
                        const u32 ba = s_ba[p0 + i];
                        const u32 at = bkt_get(ba >> 4) + (ba & 15u);
                        s_key[at] = kk[i];
                        s_pos[at] = (u32)(p0 + i);
                    }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < VT; i++)
                    if (p0 + i < (int)n) {
                        const u32 b = (u32)s_ba[p0 + i] >> 4;
                        const u32 beg = bkt_get(b), end = bkt_get(b + 1);
                        u32 before = 0;
                        for (u32 j = beg; j < end; j++) {
                            const u64 kj = s_key[j];
                            const u32 pj = s_pos[j];
                            before += (kj < kk[i] || (kj == kk[i] && pj < (u32)(p0 + i))) ? 1u : 0u;
                        }
                        s_ba[p0 + i] = (unsigned short)(beg + before);
                    }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < VT; i++)
                    if (p0 + i < (int)n) {
                        const u32 fin = s_ba[p0 + i];
                        s_key[fin] = kk[i];
                        if (TAX) s_tax[fin] = tt[i];
                    }
                __syncthreads();
                placed = true;
            }
        }
        // ---- a thread's VT consecutive tile positions in stable order: rank = #smaller + #equal in front ------------------
        if (!placed) {
        {
            u64 kk[VT];
            u32 tt[VT];
            int tr = tid;
            asm volatile("" : "+v"(tr));  // index math of a phase is recomputed, not kept live across the pass loop
            const int p0 = tr * VT;
#pragma unroll
            for (int i = 0; i < VT; i++) {
                const bool valid = p0 + i < (int)n;
                kk[i] = valid ? s_key[p0 + i] : SR_MAX;
                if (TAX) tt[i] = valid ? s_tax[p0 + i] : 0u;
            }
            u32 rk[VT];
#pragma unroll
            for (int i = 0; i < VT; i++) rk[i] = 0;
#pragma unroll
            for (int i = 0; i < VT; i++)
#pragma unroll
                for (int q = i + 1; q < VT; q++) {
                    const bool c = kk[q] < kk[i];
                    rk[i] += c ? 1u : 0u;
                    rk[q] += c ? 0u : 1u;
                }
            __syncthreads();  // every thread has read its positions: the tile is rewritten as runs of VT with sentinels
            const int base = tr * (VT + 1);
#pragma unroll
            for (int i = 0; i < VT; i++) {
                s_key[base + (int)rk[i]] = kk[i];
                if (TAX) s_tax[base + (int)rk[i]] = tt[i];
            }
            s_key[base + VT] = SR_MAX;
        }
        danger = __syncthreads_or(danger ? 1 : 0) != 0;
        SR_PH(2);

        // ---- merge rounds: runs of VT << (k - 1) records, run i at i * (run length + 1), a sentinel behind it ----------
#ifndef SR_ABL_ROUNDS
#define SR_ABL_ROUNDS LOGNT  /* experiment only: fewer merge rounds (wrong results) */
#endif
#pragma unroll
        for (int k = 1; k <= SR_ABL_ROUNDS; k++) {
            const int L = VT << (k - 1);
            if ((int)n > L) {  // (uniform) otherwise everything already is one run
                int te = tid;
                asm volatile("" : "+v"(te));
                const int pair = te >> k, lt = te & ((1 << k) - 1);
                const int abase = 2 * pair * (L + 1), bbase = abase + L + 1;
                int la = (int)n - 2 * pair * L, lb = (int)n - (2 * pair + 1) * L;
                la = la < 0 ? 0 : (la > L ? L : la);
                lb = lb < 0 ? 0 : (lb > L ? L : lb);
                u64 ro[VT];
                u32 rt[VT];
                if (danger) sr_merge_checked<TAX, VT>(s_key, s_tax, abase, la, bbase, lb, lt, ro, rt);
                else sr_merge_fast<TAX, VT>(s_key, s_tax, abase, la, bbase, lb, lt, ro, rt);
                __syncthreads();  // every thread has read its inputs: the runs are rewritten in place
                const int obase = pair * (2 * L + 1), o0 = obase + lt * VT, Lo = la + lb;
#pragma unroll
                for (int q = 0; q < VT; q++) {
                    s_key[o0 + q] = ro[q];
                    if (TAX) s_tax[o0 + q] = rt[q];
                }
                if ((Lo >= lt * VT && Lo < lt * VT + VT) || (lt == (1 << k) - 1 && Lo == (VT << k))) s_key[obase + Lo] = SR_MAX;
                __syncthreads();
            }
        }
        }  // (!placed)

        SR_PH(3);
        // ---- emit: the merged tile is s_key / s_tax [0, n) ---------------------------------------------------------------
        int count = (int)n;
        if (UNION) {
            int tf = tid;
            asm volatile("" : "+v"(tf));
            const int i0 = tf * VT;
            u64 hk[VT];
            u32 ht[VT];
            u32 headm = 0, multim = 0, neqm = 0, passm = 0;
            const bool counted = p.threshold > 1;  // (uniform) `common`: a run is emitted when it has `threshold` records
            const int need = counted ? (int)p.threshold - 1 : 0;
            {
                u64 prevk = (i0 > 0 && i0 <= (int)n) ? s_key[i0 - 1] : 0;
                u32 prevt = (TAX && i0 > 0 && i0 <= (int)n) ? s_tax[i0 - 1] : 0;
                u64 kcur = s_key[i0];
#pragma unroll
                for (int s = 0; s < VT; s++) {
                    const int i = i0 + s;
                    const u64 knext = s_key[i + 1];  // (inside the buffer; garbage behind the tile is never used)
                    const bool in = i < (int)n;
                    const bool head = in && (i == 0 || kcur != prevk);
                    hk[s] = kcur;
                    headm |= head ? (1u << s) : 0u;
                    // (the tile is sorted: the run is that long iff the record `need` places on carries the same code)
                    if (counted && head && i + need < (int)n && s_key[i + need] == kcur) passm |= 1u << s;
                    if (TAX) {
                        const u32 t = s_tax[i];
                        ht[s] = t;
                        const bool follows = in && !head;
                        const bool multi = follows || (in && i + 1 < (int)n && knext == kcur);
                        multim |= multi ? (1u << s) : 0u;
                        neqm |= (follows && t != prevt) ? (1u << s) : 0u;
                        prevt = t;
                    }
                    prevk = kcur;
                    kcur = knext;
                }
            }
            if (counted && !TAX) headm = passm;  // plain codes: the runs that fall short simply are no heads
            u32 tot;
            const u32 hexcl = block_excl_scan_u32<NT>((u32)__popc(headm), s_scan, &tot);
            // (the scan's barriers lie between every thread's last read of the tile and the writes below)
            // (Measured and dropped: runs of two or three records folded by their head with table LCAs straight away instead
            //  of through the pre-order numbers -- 1000 files that hardly overlap: emit phase 65 k -> 97 k cycles per range;
            //  a thread's serial chain of dependent table reads is worse than independent look-ups plus LDS atomics.)
            if (TAX) {
                u64 *s_acc = s_key;   // per output slot: [31:0] smallest, [63:32] largest pre-order number of the run
                u32 *s_flag = s_tax;  // per output slot: SR_NEQ | SR_BAD
                if (!counted) {
                    for (u32 w = (u32)tid; w < tot; w += NT) {
                        s_acc[w] = 0x00000000FFFFFFFFull;
                        s_flag[w] = 0;
                    }
                } else {
                    // every head prepares its own slot and says whether its run counts; the records of a run that falls
                    // short skip the fold (no pre-order look-up, no atomics, no table LCA)
                    u32 w = hexcl;
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        if (headm & (1u << s)) {
                            s_acc[w] = 0x00000000FFFFFFFFull;
                            s_flag[w] = (passm & (1u << s)) ? 0u : (u32)SR_DEAD;
                            w++;
                        }
                    }
                }
                __syncthreads();
                if (counted && multim) {
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        if (multim & (1u << s)) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            if (s_flag[w] & SR_DEAD) multim &= ~(1u << s);
                        }
                    }
                }
                if (p.clade_emit) {
                    // Round 5, taxids of unrelated taxa (the host's sample: SrArgs::clade_emit): a record brings the one-byte
                    // CLADE code of its taxid (a 2.4 MB table that stays in L2) instead of its 4-byte pre-order number; the
                    // run's two words hold `code << 24 | number` with a sentinel number (all ones in the minimum, zero in
                    // the maximum).  A run whose taxids span two clades has the LCA of those two clade nodes -- one read of the
                    // clade-pair table per head, no node_at / root-path reads; runs that stayed inside ONE clade are marked,
                    // and only their records fetch their numbers in a second, sparse pass and fold them exactly.
                    u32 en[VT];
#pragma unroll
                    for (int s = 0; s < VT; s++) {  // (all of the thread's table reads in flight together; clade8[0] = 0)
                        const u32 t = ht[s];
                        en[s] = (u32)p.tax.clade8[((multim & (1u << s)) && t < p.tax.size) ? t : 0u] << 24;
                    }
                    {
                        u32 cw = 0xFFFFFFFFu, cmn = 0xFFFFFFFFu, cmx = 0u, cfl = 0u;
                        auto flush = [&]() {
                            if (cw == 0xFFFFFFFFu) return;
                            if (cmx) {
                                atomicMin(reinterpret_cast<u32 *>(&s_acc[cw]), cmn);
                                atomicMax(reinterpret_cast<u32 *>(&s_acc[cw]) + 1, cmx);
                            }
                            if (cfl) atomicOr(&s_flag[cw], cfl);
                        };
#pragma unroll
                        for (int s = 0; s < VT; s++) {
                            if (multim & (1u << s)) {
                                const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                                if (w != cw) {
                                    flush();
                                    cw = w; cmn = 0xFFFFFFFFu; cmx = 0u; cfl = 0u;
                                }
                                const u32 eu = en[s];
                                cfl |= (neqm & (1u << s)) ? (u32)SR_NEQ : 0u;
                                if (eu == 0) cfl |= (u32)SR_BAD;
                                else {
                                    cmn = (eu | 0xFFFFFFu) < cmn ? (eu | 0xFFFFFFu) : cmn;
                                    cmx = eu > cmx ? eu : cmx;
                                }
                            }
                        }
                        flush();
                    }
                    __syncthreads();
                    // heads: which runs stayed inside one clade?
                    bool any_exact = false;
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        if ((headm & multim) & (1u << s)) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            const u32 fl = s_flag[w];
                            const u64 acc = s_acc[w];
                            if ((fl & SR_NEQ) && !(fl & SR_BAD) && ((u32)acc >> 24) == ((u32)(acc >> 32) >> 24)) {
                                s_flag[w] = fl | (u32)SR_EXACT;  // (its only writer now: every fold lies in front of the barrier)
                                any_exact = true;
                            }
                        }
                    }
                    if (__syncthreads_or(any_exact ? 1 : 0)) {
                        u32 ex[VT];
                        u32 xm = 0;
#pragma unroll
                        for (int s = 0; s < VT; s++) {
                            bool x = false;
                            if (multim & (1u << s)) {
                                const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                                x = (s_flag[w] & SR_EXACT) != 0 && en[s] != 0;
                            }
                            xm |= x ? (1u << s) : 0u;
                            ex[s] = p.tax.euler[x ? ht[s] : 0u];
                        }
#pragma unroll
                        for (int s = 0; s < VT; s++) {
                            if (xm & (1u << s)) {
                                const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                                const u32 e = en[s] | ex[s];
                                atomicMin(reinterpret_cast<u32 *>(&s_acc[w]), e);
                                atomicMax(reinterpret_cast<u32 *>(&s_acc[w]) + 1, e);
                            }
                        }
                        __syncthreads();
                    }
                    // one read of the clade-pair table per head whose run spans two clades (all of a thread's in flight)
                    u32 qi[VT];
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        qi[s] = 0;
                        if ((headm & multim) & (1u << s)) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            const u32 fl = s_flag[w];
                            if ((fl & SR_NEQ) && !(fl & (SR_BAD | SR_EXACT))) {
                                const u64 acc = s_acc[w];
                                qi[s] = ((u32)acc >> 24) * p.tax.kp + ((u32)(acc >> 32) >> 24);
                            }
                        }
                    }
                    u32 qv[VT];
#pragma unroll
                    for (int s = 0; s < VT; s++) qv[s] = p.tax.pair[qi[s]];
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        if ((headm & multim) & (1u << s)) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            const u32 fl = s_flag[w];
                            if (fl & SR_NEQ) {
                                if (fl & SR_BAD) ht[s] = 0;
                                else if (fl & SR_EXACT) {
                                    const u64 acc = s_acc[w];
                                    ht[s] = lca_dev(p.tax, p.tax.node_at[(u32)acc & 0xFFFFFFu], p.tax.node_at[(u32)(acc >> 32) & 0xFFFFFFu]);
                                } else ht[s] = qv[s];
                            }
                        }
                    }
                    __syncthreads();
                } else {
                if (multim) {
                    // a thread's consecutive records of one run are folded in registers first: one pair of LDS atomics per
                    // (thread, run) instead of per record (a code that is in 900 files puts 900 atomics on one address)
                    u32 en[VT];
#pragma unroll
                    for (int s = 0; s < VT; s++) {  // (all of the thread's table reads in flight together; euler[0] = 0)
                        const u32 t = ht[s];
                        en[s] = p.tax.euler[((multim & (1u << s)) && t < p.tax.size) ? t : 0u];
                    }
                    u32 cw = 0xFFFFFFFFu, cmn = 0xFFFFFFFFu, cmx = 0u, cfl = 0u;
                    auto flush = [&]() {
                        if (cw == 0xFFFFFFFFu) return;
                        if (cmx) {
                            atomicMin(reinterpret_cast<u32 *>(&s_acc[cw]), cmn);
                            atomicMax(reinterpret_cast<u32 *>(&s_acc[cw]) + 1, cmx);
                        }
                        if (cfl) atomicOr(&s_flag[cw], cfl);
                    };
#pragma unroll
                    for (int s = 0; s < VT; s++) {
                        if (multim & (1u << s)) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            if (w != cw) {
                                flush();
                                cw = w; cmn = 0xFFFFFFFFu; cmx = 0u; cfl = 0u;
                            }
                            const u32 eu = en[s];
                            cfl |= (neqm & (1u << s)) ? (u32)SR_NEQ : 0u;
                            if (eu == 0) cfl |= (u32)SR_BAD;
                            else {
                                cmn = eu < cmn ? eu : cmn;
                                cmx = eu > cmx ? eu : cmx;
                            }
                        }
                    }
                    flush();
                }
                __syncthreads();
                // one table LCA per run head, THREE heads of a thread at a time with their reads in flight together: the two
                // node_at reads of each, then the first root-path chunks of each pair (lca_begin), then the comparisons.  One
                // head after the other was a chain of two dependent table reads per head, up to eighteen per thread
                // (files that hardly overlap: most records are heads of runs of two or three).
#pragma unroll
                for (int g0 = 0; g0 < VT; g0 += 3) {
                    u32 na[3], nb[3];
                    bool need[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const int s = g0 + q;
                        need[q] = false;
                        na[q] = nb[q] = 0;
                        if (s < VT && ((headm & multim) & (1u << s))) {
                            const u32 w = hexcl + (u32)__popc(headm & ((2u << s) - 1u)) - 1u;
                            const u32 fl = s_flag[w];
                            if (fl & SR_NEQ) {
                                if (fl & SR_BAD) ht[s] = 0;
                                else {
                                    const u64 acc = s_acc[w];
                                    need[q] = true;
                                    na[q] = (u32)acc;
                                    nb[q] = (u32)(acc >> 32);
                                }
                            }
                        }
                    }
                    if (!(need[0] | need[1] | need[2])) continue;
#pragma unroll
                    for (int q = 0; q < 3; q++) {  // (node_at[0] is mapped: numbers start at 1)
                        na[q] = p.tax.node_at[need[q] ? na[q] : 0u];
                        nb[q] = p.tax.node_at[need[q] ? nb[q] : 0u];
                    }
                    LcaReq rq[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) lca_begin(p.tax, need[q] ? na[q] : 0u, need[q] ? nb[q] : 0u, rq[q]);
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        if (g0 + q < VT && need[q]) ht[g0 + q] = lca_finish(p.tax, rq[q]);
                }
                __syncthreads();
                }  // (!clade_emit)
            }
            u32 w = hexcl;
            if (counted && TAX) {  // the slots were per run: the runs that count are packed once more
                headm = passm;
                w = block_excl_scan_u32<NT>((u32)__popc(headm), s_scan, &tot);
            }
#pragma unroll
            for (int s = 0; s < VT; s++) {
                if (headm & (1u << s)) {
                    s_key[w] = hk[s];
                    if (TAX) s_tax[w] = ht[s];
                    w++;
                }
            }
            __syncthreads();
            count = (int)tot;
        }
        {
            u64 *o = p.out_keys + out_pos;
            for (int i = tid; i < count; i += NT) o[i] = s_key[i];
            if (TAX) {
                u32 *to = p.out_tax + out_pos;
                for (int i = tid; i < count; i += NT) to[i] = s_tax[i];
            }
            out_pos += (u64)count;
        }
        SR_PH(4);
#ifdef SR_PHASES
        if (tid == 0) atomicAdd((unsigned long long *)&p.result[13], 1ull);
#endif
        if (!more) break;
        __syncthreads();  // the tile is rewritten by the next pass
#pragma unroll
        for (int q = 0; q < SR_SPT; q++) lo[q] = e[q];
    }
    if (UNION && tid == 0) {
        p.out_cnt[r] = out_pos - out_pos0;
        p.out_off[r] = out_pos0;
    }
}

// gather the ranges' slots of a union into the caller's buffer: range r goes to dst + excl[r]
__global__ void sr_gather_kernel(const u64 *src, const u32 *tsrc, const u64 *slot, const u64 *cnt, const u64 *excl, u64 *dst,
                                 u32 *tdst, u64 cap) {
    const u32 r = blockIdx.x;
    const u64 off = slot[r], n = cnt[r], d0 = excl[r];
    if (d0 + n > cap) return;  // the host reports UKM_ERR_CAPACITY
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) dst[d0 + i] = src[off + i];
    if (tsrc)
        for (u64 i = threadIdx.x; i < n; i += blockDim.x) tdst[d0 + i] = tsrc[off + i];
}

template <bool TAX, bool UNION, int NT, int VT, int LOGNT>
int sr_launch(const SrArgs &a, hipStream_t st) {
    if (a.buckets == 1) hipLaunchKernelGGL((sr_merge_kernel<TAX, UNION, NT, VT, LOGNT, 1>), dim3(a.per_xcd * 8), dim3(NT), 0, st, a);
    else if (a.buckets == 2) hipLaunchKernelGGL((sr_merge_kernel<TAX, UNION, NT, VT, LOGNT, 2>), dim3(a.per_xcd * 8), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((sr_merge_kernel<TAX, UNION, NT, VT, LOGNT, 0>), dim3(a.per_xcd * 8), dim3(NT), 0, st, a);
    return UKM_OK;
}

#ifndef SR_NT
#define SR_NT 512
#endif
#ifndef SR_VT
#define SR_VT 9
#endif
constexpr int SR_LOGNT = SR_NT == 512 ? 9 : (SR_NT == 256 ? 8 : 10);
constexpr int SR_CAP = SR_NT * SR_VT;

}  // namespace

int ukm_srmerge_mode(const ukm_ctx *c) { return ukm_env_int(c, "UKM_SRMERGE", -1); }

int ukm_dev_srmerge(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                    u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, u32 threshold) {
    *fallback = true;
    *n_out = 0;
    const bool uni = op == UKM_KWAY_UNION;
    const int mode = ukm_srmerge_mode(c);
    if (mode == 0 || S < 2 || S > SR_MAX_STREAMS) return UKM_OK;
    u64 N = 0;
    for (int j = 0; j < S; j++) {
        if (lens[j] == 0 || lens[j] >= (1ull << 32)) return UKM_OK;  // (callers drop empty streams)
        N += lens[j];
    }
    if (mode < 1) {
        // The library's own choice, from measurements against the multi-level merge (profiles/r04_notes.md, 1e9 records):
        // this route costs the same per record whatever the number of streams (nine merge rounds per tile), the levels
        // cost one pass per factor of eight.  At 1000 streams x 1e6 records it wins with taxids (a level moves 24 B per
        // record: merge 24.8 against 27.8 ms, union of files that overlap little 39 - 48 against 50 - 63 ms) and ties or
        // loses by a few per cent on plain codes (19.6 - 22.8 against 19.0 - 20.0 ms); at 300 streams it loses (24.2
        // against 21.7 ms), at 100 by far.
        // `common` below the number of files (threshold > 1) otherwise is the whole merge plus a counting scan of it.
        if ((!tax && threshold <= 1) || S < 512 || N < (1ull << 24)) return UKM_OK;
    }
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "merge: taxids given but out_taxids is NULL");
    if (tax && uni && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "union: records carry taxids but no taxonomy is loaded");
    if (tax && uni && (c->tax_euler == nullptr || c->tax_node_at == nullptr)) return UKM_OK;
    if (!uni && N > out_cap) {
        *n_out = N;
        UKM_FAIL(UKM_ERR_CAPACITY, "merge: output needs %llu records, capacity is %llu", (unsigned long long)N,
                 (unsigned long long)out_cap);
    }
    // ---- ranges: ~85 % of a tile on average, splitters from 128 samples per range -------------------------------------
    const int fill_pct = ukm_env(c, "UKM_SRMERGE_FILL") ? std::max(10, std::min(400, atoi(ukm_env(c, "UKM_SRMERGE_FILL")))) : 85;  // developer knob
    const u64 target = std::max<u64>(1, (u64)SR_CAP * (u64)fill_pct / 100);
    u64 R64 = (N + target - 1) / target;
    u64 D = 1, ns = 0;
    std::vector<u64> sample_base((size_t)S + 1, 0);
    if (R64 > 1) {
        const u64 spr = ukm_env(c, "UKM_SRMERGE_SPR") ? std::max(8, atoi(ukm_env(c, "UKM_SRMERGE_SPR"))) : SR_SAMPLES_PER_RANGE;  // developer knob
        D = std::max<u64>(1, N / (R64 * spr));
        for (int j = 0; j < S; j++) sample_base[(size_t)j + 1] = sample_base[(size_t)j] + lens[j] / D;
        ns = sample_base[(size_t)S];
        if (ns < R64) R64 = std::max<u64>(1, ns);
    }
    // cut tables of more than 4 GB each (or a splitter index that overflows): the multi-level merge answers
    if (R64 >= (1ull << 28) || (u64)S * (R64 + 1) > (1ull << 30) || ns >= (1ull << 34)) return UKM_OK;
    const u32 R = (u32)R64;
    const u32 RP = R + 1;

    const bool dbg = ukm_env(c, "UKM_SRMERGE_DEBUG") != nullptr;
    std::vector<std::pair<const char *, hipEvent_t>> marks;
    auto mark = [&](const char *name) {
        if (!dbg) return;
        hipEvent_t e;
        if (hipEventCreate(&e) == hipSuccess) {
            (void)hipEventRecord(e, c->stream);
            marks.emplace_back(name, e);
        }
    };
    auto drop_marks = [&]() {
        for (auto &m : marks) (void)hipEventDestroy(m.second);
        marks.clear();
    };
    mark("start");

    // ---- device tables: [keys S][tax S][len S][sample_base S + 1][seg_base S + 1] ------------------------------------
    const size_t ntab = (size_t)5 * S + 2;
    std::vector<u64> tab(ntab);
    u64 nseg = 0;
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        tab[(size_t)2 * S + j] = lens[j];
        tab[(size_t)4 * S + 1 + j] = nseg;
        nseg += (lens[j] + (u64)CT * CT_SEG - 1) / ((u64)CT * CT_SEG);
    }
    tab[(size_t)5 * S + 1] = nseg;
    for (int j = 0; j <= S; j++) tab[(size_t)3 * S + j] = sample_base[(size_t)j];
    if (nseg > 0x7FFFFFFFull) return UKM_OK;
    u64 *d_tab = nullptr;
    UKM_TRY(ws_alloc_t(c, ntab, &d_tab));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), ntab * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    const u64 *const *d_keys = reinterpret_cast<const u64 *const *>(d_tab);
    const u32 *const *d_tax = reinterpret_cast<const u32 *const *>(d_tab + S);
    const u64 *d_len = d_tab + 2 * (size_t)S;
    const u64 *d_sbase = d_tab + 3 * (size_t)S;
    const u64 *d_segbase = d_tab + 4 * (size_t)S + 1;

    u64 *ctl = nullptr, *spl = nullptr;
    u64 sample_dups = 0;
    u32 *cuts = nullptr;
    UKM_TRY(ws_alloc_t(c, 32, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, 32 * sizeof(u64), c->stream));
    UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &spl));
    UKM_TRY(ws_alloc_t(c, (size_t)S * RP, &cuts));
    if (R > 1) {
        u64 *samples = nullptr;
        UKM_TRY(ws_alloc_t(c, ns, &samples));
        hipLaunchKernelGGL(sr_sample_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, d_keys, d_sbase, (u32)S, D,
                           ns, samples);
        UKM_TRY(ukm_dev_sort(c, samples, nullptr, ns, 64));
        hipLaunchKernelGGL(sr_splitters_kernel, dim3((R + 255) / 256), dim3(256), 0, c->stream, samples, ns, R, spl);
        hipLaunchKernelGGL(sr_dupcount_kernel, dim3((unsigned)(ns / 256 + 1 < 2048 ? ns / 256 + 1 : 2048)), dim3(256), 0, c->stream, samples, ns, ctl + 2);
    }
    mark("sample+sort");
    {
        CutArgs a;
        a.leaf_keys = d_keys;
        a.leaf_len = d_len;
        a.seg_base = d_segbase;
        a.spl = spl;
        a.cuts = cuts;
        a.result = ctl;
        a.S = (u32)S;
        a.R = R;
        hipLaunchKernelGGL(sr_cuts_kernel, dim3((unsigned)nseg), dim3(CT_NT), 0, c->stream, a);
        UKM_HIP(hipGetLastError());
    }
    mark("cuts");
    const TaxDev taxd = ukm_taxdev(c);
    const bool clade_able = uni && tax && taxids && taxd.clade8 != nullptr && taxd.pair != nullptr && taxd.euler != nullptr && S >= 2;
    if (clade_able) {  // (rides on the read-back below)
        hipLaunchKernelGGL(sr_taxsample_kernel, dim3(16), dim3(256), 0, c->stream, d_tax, d_len, (u32)S, taxd, ctl + 4);
        UKM_HIP(hipGetLastError());
    }
    u64 tax_pairs = 0, tax_same = 0;
    {
        u64 fl2[5] = {0, 0, 0, 0, 0};
        UKM_TRY(ukm_read_u64(c, ctl + 1, fl2, 5));
        tax_pairs = fl2[3];
        tax_same = fl2[4];
        const u64 fl = fl2[0];
        // Counting placement or merge rounds?  A code with c copies among the streams shows up ~(c - 1) / (2 D) times as
        // an equal neighbour in the sorted sample (every D-th record).  Placement wins while tiles hold next to no equal
        // codes (1000 files x 1e6 with taxids, merge pass: 1.6 copies per code 19.6 against 24.3 ms, 4 copies 23.5 against
        // 23.9) and loses its counting phase when buckets overflow (20 copies: 22.5 against 19.7): up to ~2.5 copies.
        sample_dups = fl2[1];
        if (fl & SR_FLAG_UNSORTED) {
            drop_marks();
            return UKM_OK;  // *fallback: the caller's general route sorts / reports it
        }
    }
    // ---- the merge pass -----------------------------------------------------------------------------------------------
    u64 *slots_k = nullptr, *cnt = nullptr, *slot_off = nullptr;
    u32 *slots_t = nullptr;
    if (uni) {
        UKM_TRY(ws_alloc_t(c, N + 1, &slots_k));
        if (tax) UKM_TRY(ws_alloc_t(c, N + 1, &slots_t));
        UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &cnt));
        UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &slot_off));
    }
    SrArgs a;
    memset(&a, 0, sizeof(a));
    a.leaf_keys = d_keys;
    a.leaf_tax = tax ? d_tax : nullptr;
    a.cuts = cuts;
    a.spl = spl;
    a.out_keys = uni ? slots_k : out;
    a.out_tax = tax ? (uni ? slots_t : tout) : nullptr;
    a.out_cnt = cnt;
    a.out_off = slot_off;
    a.result = ctl;
    a.S = (u32)S;
    a.R = R;
    a.per_xcd = (R + 7) / 8;
    a.threshold = uni ? threshold : 0;
    {
        const int force = ukm_env_int(c, "UKM_SRMERGE_BUCKETS", -1);  // developer knob
        const double dup_share = ns ? (double)sample_dups / (double)ns : 0.0;
        const double extra = dup_share * 2.0 * (double)D;  // ~copies per code - 1
        // dense numbers: a tile of SR_CAP records must hold at most SR_DMAX distinct codes, i.e. >= 4.5 copies per code (the sample underestimates: 1000 files x 1e6 with taxids: 5.8 estimated copies 29.6 -> 27.6 ms, 3.7 estimated copies slower)
        a.buckets = force >= 0 ? (u32)force : (R > 1 && extra < 1.5 ? 1u : (R > 1 && extra >= 4.3 ? 2u : 0u));
        if (a.buckets > 2) a.buckets = 0;
        if (dbg) fprintf(stderr, "[srmerge] %llu of %llu samples equal their predecessor: ~%.1f copies per code -> %s\n", (unsigned long long)sample_dups,
                         (unsigned long long)ns, 1.0 + extra,
                         a.buckets == 1 ? "counting placement by value" : (a.buckets == 2 ? "counting placement by dense code numbers" : "merge rounds"));
    }
    a.tax = taxd;
    {
        // taxids of unrelated taxa: the emit folds clade codes (one byte per record) and fetches a number only for the runs
        // that stay inside one clade; related taxa would send most runs through that second pass (UKM_SRMERGE_CLADE = 0 / 1)
        const int k = ukm_env_int(c, "UKM_SRMERGE_CLADE", -1);
        a.clade_emit = (clade_able && k != 0 && (k == 1 || (tax_pairs >= 256 && tax_same * 8 < tax_pairs))) ? 1u : 0u;
        if (dbg) fprintf(stderr, "[srmerge] taxid sample: %llu of %llu pairs in one clade with different taxids: clade emit %u\n",
                         (unsigned long long)tax_same, (unsigned long long)tax_pairs, a.clade_emit);
    }
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (tax) {
        if (uni) sr_launch<true, true, SR_NT, SR_VT, SR_LOGNT>(a, c->stream);
        else sr_launch<true, false, SR_NT, SR_VT, SR_LOGNT>(a, c->stream);
    } else {
        if (uni) sr_launch<false, true, SR_NT, SR_VT, SR_LOGNT>(a, c->stream);
        else sr_launch<false, false, SR_NT, SR_VT, SR_LOGNT>(a, c->stream);
    }
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    mark("merge");
    if (uni) {
        u64 *excl = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &excl));
        UKM_TRY(ukm_dev_exclusive_scan_u64(c, cnt, excl, R, ctl));  // ctl[0] = total
        hipLaunchKernelGGL(sr_gather_kernel, dim3(R), dim3(256), 0, c->stream, slots_k, tax ? slots_t : nullptr, slot_off, cnt, excl, out,
                           tout, out_cap);
        UKM_HIP(hipGetLastError());
        mark("gather");
    }
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    if (dbg) {
        fprintf(stderr, "[srmerge] S=%d N=%llu R=%u ns=%llu D=%llu flags=%llu out=%llu :", S, (unsigned long long)N, R,
                (unsigned long long)ns, (unsigned long long)D, (unsigned long long)h[1], (unsigned long long)(uni ? h[0] : N));
        for (size_t i = 1; i < marks.size(); i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
            fprintf(stderr, " %s=%.3fms", marks[i].first, ms);
        }
        fprintf(stderr, "\n");
#ifdef SR_PHASES
        u64 ph[16];
        UKM_TRY(ukm_read_u64(c, ctl + 8, ph, 16));
        fprintf(stderr, "[srmerge] k-cycles per range: prologue %.1f gather %.1f rank %.1f rounds %.1f emit %.1f\n", ph[0] / 1e3 / R,
                ph[1] / 1e3 / R, ph[2] / 1e3 / R, ph[3] / 1e3 / R, ph[4] / 1e3 / R);
        fprintf(stderr, "[srmerge] passes %llu, of them by the quota rule %llu; ranges over one tile %llu, over two %llu, largest %llu\n",
                (unsigned long long)ph[5], (unsigned long long)ph[6], (unsigned long long)ph[9], (unsigned long long)ph[10], (unsigned long long)ph[8]);
#endif
    }
    drop_marks();
    if (h[1] & (SR_FLAG_UNSORTED | SR_FLAG_DEGENERATE)) return UKM_OK;  // *fallback stays set
    *fallback = false;
    *n_out = uni ? h[0] : N;
    if (*n_out > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "union: output needs %llu records, capacity is %llu", (unsigned long long)*n_out,
                 (unsigned long long)out_cap);
    return UKM_OK;
}
