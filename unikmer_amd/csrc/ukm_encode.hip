// ukm_encode.hip — sequence -> uint64 window values on the GPU:
//   * 2-bit k-mer codes with optional canonicalisation: replaces
//     sketches.NewKmerIterator(...).NextKmer() (count.go:321,363) = shenwei356/kmers v0.1.0
//     (A=0 C=1 G=2 T/U=3, first base most significant, canonical = min(code, revcomp));
//   * ntHash v1 (will-rowe/nthash v0.4.0): replaces sketches.NewHashIterator(...).NextHash()
//     (count.go:319,361) / nthash.Hasher.Next (dump.go:253-260), fused with the Scaled-MinHash
//     filter `code > maxHash -> skip` (count.go:98,373-375).
//
// Kernel shape: one 256-thread workgroup per tile of 4096 consecutive base positions of the
// concatenated records.  The tile (+k-1 bases of overlap) is loaded coalesced into LDS once;
// each thread then ROLLS over a strip of 16 consecutive windows (k-1 warm-up steps, then one
// base per window), so a window costs ~3 base steps instead of k.  Record boundaries only
// decide validity and the output index (out_off[r] + p - rec_off[r]); the rolling state is
// boundary-agnostic.  Windows that wrap (circular genomes) are recomputed directly.
// Algorithmic bytes: 1 B/base read, 8 B/window written (8/scale with the Scaled filter).
#include <algorithm>

#include "ukm_device.h"

namespace {

constexpr int NT = 256;
constexpr int WPT = 16;            // windows per thread
constexpr int TB = NT * WPT;       // base positions per tile
constexpr int SB = TB + 64;        // LDS bytes for bases (k <= 64)

// kmers v0.1.0 base table; 4 = illegal base
__device__ __forceinline__ u32 base2bit(u32 c) {
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': case 'M': case 'm': case 'V': case 'v':
    case 'H': case 'h': case 'R': case 'r': case 'D': case 'd': case 'W': case 'w': return 0;
    case 'C': case 'c': case 'S': case 's': case 'B': case 'b': case 'Y': case 'y': return 1;
    case 'G': case 'g': case 'K': case 'k': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

__device__ __forceinline__ u64 nt_seed(u32 c) {
    switch (c) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': case 'U': case 'u': return SEED_T;
    default: return 0;
    }
}
__device__ __forceinline__ u64 nt_cseed(u32 c) {
    switch (c) {
    case 'A': case 'a': return SEED_T;
    case 'C': case 'c': return SEED_G;
    case 'G': case 'g': return SEED_C;
    case 'T': case 't': case 'U': case 'u': return SEED_A;
    default: return 0;
    }
}
__device__ __forceinline__ u64 rol64(u64 x, u32 s) { s &= 63; return s ? (x << s) | (x >> (64 - s)) : x; }
__device__ __forceinline__ u64 ror64(u64 x, u32 s) { s &= 63; return s ? (x >> s) | (x << (64 - s)) : x; }

struct WinArgs {
    const u8 *bases;
    const u64 *rec_off;  // [n_rec + 1]
    const u64 *out_off;  // [n_rec] exclusive scan of per-record window counts
    u64 n_rec;
    u64 total_bases;
    int k;
    int canonical;
    int circular;
    u64 max_hash;
    u64 *out;
    u64 out_cap;
    u64 *status;  // FILTER only
    u32 *ticket;  // FILTER only
    u64 *result;  // [0] total (FILTER), [1] flags: bit0 = illegal base inside an emitted window
    u64 ntiles;
};

__global__ void window_count_kernel(const u64 *rec_off, u64 n_rec, int k, int circular, u64 *cnt) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    u64 len = rec_off[r + 1] - rec_off[r];
    cnt[r] = (len < (u64)k) ? 0 : (circular ? len : len - (u64)k + 1);
}

// first index i in [lo, hi) with a[i] > x
__device__ __forceinline__ u64 upper_bound_u64(const u64 *a, u64 lo, u64 hi, u64 x) {
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// HASH: false = 2-bit codes, true = ntHash.  FILTER: Scaled filter + order-preserving compaction.
template <bool HASH, bool FILTER>
__global__ __launch_bounds__(NT) void window_kernel(WinArgs p) {
    __shared__ u8 s_b[SB];
    __shared__ u64 s_r[2];
    __shared__ u32 s_scan[NT / 64 + 1];
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x;
    u64 tile;
    if (FILTER) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    } else {
        tile = blockIdx.x;
    }
    const u64 P0 = tile * (u64)TB;
    const int k = p.k;
    // stage bases [P0, P0 + TB + k - 1) into LDS
    {
        const u64 lim = p.total_bases;
        const int need = TB + k - 1;
        if ((((uintptr_t)p.bases) & 3) == 0) {
            const u32 *b4 = (const u32 *)(p.bases + P0);  // P0 is a multiple of 4096
            for (int i = tid; i * 4 < need; i += NT) {
                u64 g = P0 + (u64)i * 4;
                u32 w = 0;
                if (g + 4 <= lim) w = b4[i];
                else {
                    for (int q = 0; q < 4; q++)
                        if (g + q < lim) w |= (u32)p.bases[g + q] << (8 * q);
                }
                ((u32 *)s_b)[i] = w;
            }
        } else {
            for (int i = tid; i < need; i += NT) s_b[i] = (P0 + i < lim) ? p.bases[P0 + i] : 0;
        }
    }
    if (tid == 0) s_r[0] = upper_bound_u64(p.rec_off, 0, p.n_rec + 1, P0);            // first rec_off > P0
    if (tid == 1) s_r[1] = upper_bound_u64(p.rec_off, 0, p.n_rec + 1, P0 + TB - 1);   // first rec_off > last pos
    __syncthreads();

    const u64 p_first = P0 + (u64)tid * WPT;
    // record containing p_first: last r with rec_off[r] <= p_first
    u64 r = 0;
    bool in_rec = false;
    u64 rs = 0, re = 0;  // current record [rs, re)
    if (p_first < p.total_bases && p.n_rec > 0) {
        u64 ub = upper_bound_u64(p.rec_off, s_r[0] ? s_r[0] - 1 : 0, s_r[1] < p.n_rec + 1 ? s_r[1] + 1 : p.n_rec + 1, p_first);
        if (ub > 0 && ub <= p.n_rec) {
            r = ub - 1;
            rs = p.rec_off[r];
            re = p.rec_off[r + 1];
            in_rec = true;
        }
    }

    u64 fwd = 0, rev = 0;
    const u64 mask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    int since_bad = k;  // steps since the last illegal base (saturates at k)
    u64 hv[WPT];
    u32 keep = 0, illegal = 0;

#pragma unroll 1
    for (int step = 0; step < WPT + k - 1; step++) {
        const int li = tid * WPT + step;  // LDS index of the incoming base
        const u32 c = s_b[li];
        if (HASH) {
            fwd = rol64(fwd, 1) ^ nt_seed(c);
            rev = ror64(rev, 1) ^ rol64(nt_cseed(c), (u32)(k - 1));
            if (step >= k) {
                const u32 co = s_b[li - k];
                fwd ^= rol64(nt_seed(co), (u32)k);
                rev ^= ror64(nt_cseed(co), 1);
            }
        } else {
            const u32 b = base2bit(c);
            since_bad = (b > 3) ? 0 : (since_bad < k ? since_bad + 1 : k);
            fwd = ((fwd << 2) | (u64)(b & 3)) & mask;
            rev = (rev >> 2) | ((u64)(3 - (b & 3)) << (2 * (k - 1)));
        }
        if (step >= k - 1) {
            const int w = step - (k - 1);      // window index inside the strip
            const u64 pos = p_first + (u64)w;  // global start position of the window
            // advance the record cursor
            while (in_rec && pos >= re) {
                r++;
                if (r >= p.n_rec) { in_rec = false; break; }
                rs = re;
                re = p.rec_off[r + 1];
            }
            if (in_rec && pos < p.total_bases) {
                const u64 len = re - rs;
                const bool fits = pos + (u64)k <= re;
                const bool emit = len >= (u64)k && (fits || p.circular);
                if (emit) {
                    u64 f = fwd, rv = rev;
                    bool bad = !HASH && since_bad < k;
                    if (!fits) {  // circular wrap: recompute from the record itself
                        f = 0; rv = 0; bad = false;
                        const u64 o = pos - rs;
                        for (int j = 0; j < k; j++) {
                            u64 q = o + (u64)j;
                            if (q >= len) q -= len;
                            const u32 cc = p.bases[rs + q];
                            if (HASH) {
                                f ^= rol64(nt_seed(cc), (u32)(k - 1 - j));
                                rv ^= rol64(nt_cseed(cc), (u32)j);
                            } else {
                                const u32 bb = base2bit(cc);
                                if (bb > 3) bad = true;
                                f = (f << 2) | (u64)(bb & 3);
                                rv |= (u64)(3 - (bb & 3)) << (2 * j);
                            }
                        }
                    }
                    if (bad) illegal = 1;
                    u64 v = (p.canonical && rv < f) ? rv : f;
                    if (FILTER) {
                        if (v <= p.max_hash) { keep |= 1u << w; }
                        // static indexing only: select into the unrolled slot below
#pragma unroll
                        for (int q = 0; q < WPT; q++)
                            if (q == w) hv[q] = v;
                    } else {
                        const u64 oi = p.out_off[r] + (pos - rs);
                        if (oi < p.out_cap) p.out[oi] = v;
                    }
                }
            }
        }
    }
    if (illegal) atomicOr((unsigned long long *)&p.result[1], 1ull);

    if (FILTER) {
        const u32 cnt = (u32)__popc(keep);
        u32 tile_total;
        const u32 excl = block_excl_scan_u32<NT>(cnt, s_scan, &tile_total);
        if (tid < 64) {
            u64 base = lb_lookback(p.status, tile, (u64)tile_total);
            if (tid == 0) s_misc[1] = base;
        }
        __syncthreads();
        u64 pos = s_misc[1] + excl;
#pragma unroll
        for (int q = 0; q < WPT; q++)
            if (keep & (1u << q)) {
                if (pos < p.out_cap) p.out[pos] = hv[q];
                pos++;
            }
        if (tid == 0 && tile == p.ntiles - 1) p.result[0] = s_misc[1] + tile_total;
    }
}

int run_windows(ukm_ctx *c, bool hash, const u8 *bases, const u64 *rec_off, u64 n_rec, int k,
                int canonical, int circular, u64 max_hash, u64 *out, u64 out_cap, u64 *n_out,
                u64 total_bases) {
    *n_out = 0;
    if (n_rec == 0 || total_bases == 0) return UKM_OK;
    // per-record window counts -> exclusive scan
    u64 *cnt = nullptr, *off = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, n_rec, &cnt));
    UKM_TRY(ws_alloc_t(c, n_rec, &off));
    hipLaunchKernelGGL(window_count_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, c->stream,
                       rec_off, n_rec, k, circular, cnt);
    const u64 ntiles = (total_bases + TB - 1) / TB;
    const bool filter = hash && max_hash != 0;
    const size_t nctl = 8 + (filter ? lb_status_words(ntiles) : 0);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
    UKM_TRY(ukm_dev_exclusive_scan_u64(c, cnt, off, n_rec, ctl + 3));
    u64 total_windows = 0;
    UKM_TRY(ukm_read_u64(c, ctl + 3, &total_windows));
    if (total_windows == 0) return UKM_OK;
    if (!filter && total_windows > out_cap) {
        *n_out = total_windows;
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu",
                 (unsigned long long)total_windows, (unsigned long long)out_cap);
    }
    WinArgs p;
    memset(&p, 0, sizeof(p));
    p.bases = bases; p.rec_off = rec_off; p.out_off = off; p.n_rec = n_rec;
    p.total_bases = total_bases; p.k = k; p.canonical = canonical; p.circular = circular;
    p.max_hash = max_hash; p.out = out; p.out_cap = out_cap;
    p.result = ctl; p.ticket = (u32 *)(ctl + 2); p.status = ctl + 8; p.ntiles = ntiles;
    if (!hash) hipLaunchKernelGGL((window_kernel<false, false>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
    else if (!filter) hipLaunchKernelGGL((window_kernel<true, false>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
    else hipLaunchKernelGGL((window_kernel<true, true>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
    UKM_HIP(hipGetLastError());
    u64 res[2];
    UKM_TRY(ukm_read_u64(c, ctl, res, 2));
    if (res[1] & 1) UKM_FAIL(UKM_ERR_ILLEGAL_BASE, "illegal base in sequence (kmers.ErrIllegalBase)");
    *n_out = filter ? res[0] : total_windows;
    if (*n_out > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu",
                 (unsigned long long)*n_out, (unsigned long long)out_cap);
    return UKM_OK;
}

int windows_entry(ukm_ctx *ctx, bool hash, const uint8_t *bases, const uint64_t *rec_off,
                  uint64_t n_rec, int k, int canonical, int circular, uint64_t max_hash,
                  uint64_t *out, uint64_t out_cap, uint64_t *n_out, const char *name) {
    if (!ctx || !n_out || (!out && out_cap) || (n_rec && (!rec_off || !bases)))
        UKM_FAIL(UKM_ERR_INVALID, "%s: NULL argument", name);
    if (k < 1 || k > (hash ? 64 : 32)) UKM_FAIL(UKM_ERR_K, "%s: k = %d out of range", name, k);
    *n_out = 0;
    if (n_rec == 0) return UKM_OK;
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *off = nullptr;
        UKM_TRY(ukm_in_t(ctx, rec_off, n_rec + 1, &off));
        // the total number of bases is rec_off[n_rec]; rec_off[0] must be 0
        u64 ends[1];
        u64 first = 0;
        if (ukm_is_device_ptr(rec_off)) {
            UKM_TRY(ukm_read_u64(ctx, off + n_rec, ends));
            UKM_TRY(ukm_read_u64(ctx, off, &first));
        } else {
            ends[0] = rec_off[n_rec];
            first = rec_off[0];
        }
        if (first != 0) UKM_FAIL(UKM_ERR_INVALID, "%s: rec_off[0] must be 0", name);
        const u64 total_bases = ends[0];
        const u8 *b = nullptr;
        u64 *o = nullptr;
        UKM_TRY(ukm_in_t(ctx, bases, total_bases, &b));
        UKM_TRY(ukm_out_t(ctx, out, out_cap, &o));
        int r = run_windows(ctx, hash, b, off, n_rec, k, canonical, circular, max_hash, o, out_cap, n_out, total_bases);
        ukm_out_resize(ctx, out, (r == UKM_OK ? *n_out : 0) * sizeof(u64));
        return r;
    }();
    return ukm_finish(&s, rc);
}

}  // namespace

extern "C" int ukm_encode_kmers(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off,
                                uint64_t n_rec, int k, int canonical, int circular, uint64_t *out,
                                uint64_t out_cap, uint64_t *n_out) {
    return windows_entry(ctx, false, bases, rec_off, n_rec, k, canonical, circular, 0, out, out_cap, n_out,
                         "ukm_encode_kmers");
}

extern "C" int ukm_nthash(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec,
                          int k, int canonical, int circular, uint64_t max_hash, uint64_t *out,
                          uint64_t out_cap, uint64_t *n_out) {
    return windows_entry(ctx, true, bases, rec_off, n_rec, k, canonical, circular, max_hash, out, out_cap, n_out,
                         "ukm_nthash");
}

// count.go:98  maxHash := uint64(float64(^uint64(0)) / float64(scale))
extern "C" uint64_t ukm_max_hash(uint64_t scale) {
    if (scale <= 1) return ~0ull;
    double d = 18446744073709551615.0 / (double)scale;  // float64(^uint64(0)) rounds to 2^64
    return (uint64_t)d;
}
