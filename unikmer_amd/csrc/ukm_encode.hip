// ukm_encode.hip — sequence -> uint64 window values on the GPU:
//   * 2-bit k-mer codes with optional canonicalisation: replaces
//     sketches.NewKmerIterator(...).NextKmer() (count.go:321,363) = shenwei356/kmers v0.1.0
//     (A=0 C=1 G=2 T/U=3, first base most significant, canonical = min(code, revcomp));
//   * ntHash v1 (will-rowe/nthash v0.4.0): replaces sketches.NewHashIterator(...).NextHash()
//     (count.go:319,361) / nthash.Hasher.Next (dump.go:253-260), fused with the Scaled-MinHash
//     filter `code > maxHash -> skip` (count.go:98,373-375).
//
// Kernel shape (both value kinds): one 256-thread workgroup per tile of WT = 2048 window start
// positions of the concatenated records (+64 bases of overlap), staged once in LDS.
//   phase 1 (blocked over positions): record lookup -> per-position info byte (bases left in
//            the record, "record long enough" bit) and record index; the per-position
//            ingredients of the value:
//              codes : 2-bit packed tile + illegal-base bit mask (16 bases per 32-bit word)
//              ntHash: XOR-prefix sums  P[n] = xor_{m<n} ror(seed[s_m], m),
//                                       Q[n] = xor_{m<n} rol(cseed[s_m], m)   (block XOR scan)
//   phase 2 (striped over windows, so stores are coalesced): window i is computed in O(1),
//            with no rolling dependency and no k-1 warm-up:
//              code(i)  = 2k-bit field at bit 2i of the packed tile; revcomp by bit reversal
//              fwd(i)   = rol(P[i+k] ^ P[i], i+k-1),  rev(i) = ror(Q[i+k] ^ Q[i], i)
//            (the first version rolled 16 windows per thread: (16+k-1)/16 = 4x redundant work
//            at k = 51 and 128-byte-strided stores; ntHash ran at 3.5e10 bases/s).
//   Scaled filter: survivors are compacted in window order with one wave64 ballot per round, a
//   per-tile prefix and the library's decoupled look-back.
// Windows that wrap (circular genomes) are recomputed directly from the record.
// Algorithmic bytes: 1 B/base read, 8 B/window written (8/scale with the Scaled filter).
#include <stdlib.h>

#include <algorithm>

#include "ukm_device.h"

namespace {

#ifndef ENC_NT
#define ENC_NT 512
#endif
constexpr int NT = ENC_NT;
#ifndef ENC_WT
#define ENC_WT 2048
#endif
constexpr int WT = ENC_WT;      // window start positions per tile
constexpr int TBX = WT + 64;    // base positions staged per tile (k <= 64)
constexpr int WPT = WT / NT;    // windows per thread (striped)
constexpr int PPT = (TBX + NT - 1) / NT;  // positions per thread in the blocked phase (NT * PPT >= TBX)
constexpr int NWV = NT / 64;

// kmers v0.1.0 base table; 4 = illegal base
constexpr u32 base2bit(u32 c) {
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

// a ^ b ^ c in one instruction (gfx950's three-input v_bitop3_b32, truth table 0x96)
__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// branch-free rotates: for s = 0 both shifts are by 0 and x | x = x
__device__ __forceinline__ u64 rol64(u64 x, u32 s) { return (x << (s & 63)) | (x >> ((0u - s) & 63)); }
__device__ __forceinline__ u64 ror64(u64 x, u32 s) { return (x >> (s & 63)) | (x << ((0u - s) & 63)); }

// The switches above compile to chains of exec-mask branches (measured: 2050 SALU + 1100 VALU
// instructions per wave per tile, instruction-issue bound).  Every workgroup therefore builds a
// 256-entry byte table once (one entry per thread) and the per-base work becomes LDS lookups:
// low nibble = 2-bit code (4 = illegal base), high nibble = ntHash seed index (A0 C1 G2 T3, 4 = the
// zero seed of every other byte); seeds come from two 5-entry tables (forward / complement).
constexpr u32 nt_index(u32 c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}
struct ByteTable {
    u8 v[256];
};
constexpr ByteTable make_byte_table() {
    ByteTable t{};
    for (u32 c = 0; c < 256; c++) t.v[c] = (u8)(base2bit(c) | (nt_index(c) << 4));
    return t;
}
__device__ const ByteTable g_byte_table = make_byte_table();  // built at compile time

struct BaseTables {
    u8 lut[256];
    u64 seed[8];   // [0..4] forward seeds
    u64 cseed[8];  // [0..4] seeds of the complement
};
__device__ __forceinline__ void base_tables_init(BaseTables &t, int tid) {
    static_assert(NT >= 256, "one table entry per thread");
    if (tid < 256) t.lut[tid] = g_byte_table.v[tid];
    if (tid < 5) {
        const u64 f = tid == 0 ? SEED_A : tid == 1 ? SEED_C : tid == 2 ? SEED_G : tid == 3 ? SEED_T : 0ull;
        const u64 c = tid == 0 ? SEED_T : tid == 1 ? SEED_G : tid == 2 ? SEED_C : tid == 3 ? SEED_A : 0ull;
        t.seed[tid] = f;
        t.cseed[tid] = c;
    }
}

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
    const u64 *tile_rec;  // [ntiles + 3]: record containing position min(t * WT, total_bases - 1)
};

// One thread per tile boundary: the record that contains the tile's first position.  Doing the
// two ~27-step binary searches inside the tile kernel stalled every workgroup for ~25 us.
__global__ void tile_first_rec_kernel(const u64 *rec_off, u64 n_rec, u64 total_bases, u64 ntiles, u64 tile_size,
                                      u64 *tile_rec);

// cnt[n_rec] = 0, so the exclusive scan over n_rec + 1 entries ends with the total
__global__ void window_count_kernel(const u64 *rec_off, u64 n_rec, int k, int circular, u64 *cnt) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { cnt[r] = 0; return; }
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

__global__ void tile_first_rec_kernel(const u64 *rec_off, u64 n_rec, u64 total_bases, u64 ntiles, u64 tile_size,
                                      u64 *tile_rec) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles + 3) return;
    u64 pos = t * tile_size;
    if (pos > total_bases - 1) pos = total_bases - 1;
    tile_rec[t] = upper_bound_u64(rec_off, 0, n_rec + 1, pos) - 1;
}

// 64-bit XOR inclusive scan across the wave (DPP row shifts / broadcasts, cf. wave_incl_scan_u32)
__device__ __forceinline__ u32 dpp_xor_step(u32 v, const int ctrl_id) {
    switch (ctrl_id) {
    case 0: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    case 1: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    case 2: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    case 3: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    case 4: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    default: return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    }
}
__device__ __forceinline__ u64 wave_incl_xor_scan_u64(u64 v) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
#pragma unroll
    for (int s = 0; s < 6; s++) { lo = dpp_xor_step(lo, s); hi = dpp_xor_step(hi, s); }
    return ((u64)hi << 32) | lo;
}

// reverse the 2-bit groups of the low 2k bits of ~code (revcomp of kmers v0.1.0)
__device__ __forceinline__ u64 revcomp2(u64 code, int k) {
    u64 x = ~code;
    x = ((u64)__builtin_bitreverse32((u32)x) << 32) | (u64)__builtin_bitreverse32((u32)(x >> 32));  // bit reversal
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);                     // restore pair order
    return x >> (64 - 2 * k);
}

// HASH: false = 2-bit codes, true = ntHash.  FILTER: Scaled filter + order-preserving compaction.
// TICKET (FILTER only): tile ids from an atomic counter instead of blockIdx.  The counter is ONE address
// that every workgroup hits: measured 3-4 ns per tile of pure serialisation (0.15 of 0.68 ms per 1e8 bases),
// so the default takes blockIdx and relies on in-order dispatch for look-back liveness, exactly like the
// set-op kernel (watchdog -> flag -> the host re-runs this ticketed instantiation).
template <bool HASH, bool FILTER, bool TICKET = false>
__global__ __launch_bounds__(NT) void window_kernel(WinArgs p) {
    __shared__ __attribute__((aligned(16))) u8 s_b[TBX + 16];
    __shared__ u8 s_info[TBX];                     // low 7 bits: min(bases left in record, 127); bit 7: record length >= k
    __shared__ u32 s_dl[FILTER ? 1 : TBX];         // (bases - windows) in front of the position's record, relative to s_r[2]
    __shared__ u64 s_P[HASH ? TBX + 1 : 1];        // XOR prefixes (ntHash)
    __shared__ u64 s_Q[HASH ? TBX + 1 : 1];
    __shared__ u32 s_pk[HASH ? 1 : TBX / 16 + 4];  // 2-bit packed bases, 16 per word (codes)
    __shared__ u32 s_bad[HASH ? 1 : TBX / 32 + 4]; // illegal-base bit per position (codes)
    __shared__ u64 s_wtot[2 * NWV];
    __shared__ u32 s_cnt[WPT * NWV + 1];
    __shared__ u64 s_r[3];
    __shared__ u64 s_misc[2];
    __shared__ BaseTables s_t;
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    base_tables_init(s_t, tid);  // visible after the first barrier below
    u64 tile = blockIdx.x;
    if (TICKET) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    }
    const u64 P0 = tile * (u64)WT;
    const int k = p.k;
    // ---- stage bases [P0, P0 + TBX) into LDS -----------------------------------------------------
    {
        const u64 lim = p.total_bases;
        if ((((uintptr_t)p.bases) & 3) == 0) {
            const u32 *b4 = (const u32 *)(p.bases + P0);  // P0 is a multiple of 2048
            for (int i = tid; i * 4 < TBX; i += NT) {
                const u64 g = P0 + (u64)i * 4;
                u32 w = 0;
                if (g + 4 <= lim) w = b4[i];
                else
                    for (int q = 0; q < 4; q++)
                        if (g + q < lim) w |= (u32)p.bases[g + q] << (8 * q);
                ((u32 *)s_b)[i] = w;
            }
        } else {
            for (int i = tid; i < TBX; i += NT) s_b[i] = (P0 + i < lim) ? p.bases[P0 + i] : 0;
        }
    }
    if (tid == 0) {
        s_r[0] = p.tile_rec[tile];          // record of P0
        s_r[1] = p.tile_rec[tile + 2] + 1;  // every position of the tile lies in a record <= this - 1
        // output index of a window = its base position - gap(record), gap = rec_off[r] - out_off[r]
        // (bases minus windows in front of the record; non-decreasing in r); s_r[2] = gap of the first record
        if (!FILTER) { const u64 r0 = p.tile_rec[tile]; s_r[2] = p.rec_off[r0] - p.out_off[r0]; }
    }
    if (!HASH) {
        for (int i = tid; i < TBX / 16 + 4; i += NT) s_pk[i] = 0;
        for (int i = tid; i < TBX / 32 + 4; i += NT) s_bad[i] = 0;
    }
    __syncthreads();

    // ---- phase 1: blocked over positions [tid*PPT, tid*PPT + PPT) -----------------------------------
    {
        const int m0 = tid * PPT;
        const u64 p_first = P0 + (u64)m0;
        const u64 r_first = s_r[0];
        u64 r = 0, rs = 0, re = 0;
        u32 dl = 0;
        const u64 gap0 = FILTER ? 0 : s_r[2];
        bool in_rec = false;
        if (p_first < p.total_bases && p.n_rec > 0) {
            const u64 hi = (s_r[1] + 1 < p.n_rec + 1) ? s_r[1] + 1 : p.n_rec + 1;
            const u64 ub = upper_bound_u64(p.rec_off, r_first, hi, p_first);  // a few steps, L2-resident
            if (ub > 0 && ub <= p.n_rec) {
                r = ub - 1;
                rs = p.rec_off[r];
                re = p.rec_off[r + 1];
                if (!FILTER) dl = (u32)(rs - p.out_off[r] - gap0);
                in_rec = true;
            }
        }
        u64 accP = 0, accQ = 0;
        u64 lp[PPT], lq[PPT];
        u32 pk_w0 = 0, pk_w1 = 0, bad_w0 = 0, bad_w1 = 0;  // codes only
        static_assert(PPT <= 16, "a thread's positions must span at most two packed words");
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            const int m = m0 + s;
            const u64 pos = p_first + (u64)s;
            while (in_rec && pos >= re) {  // advance the record cursor (skips empty records)
                r++;
                if (r >= p.n_rec) { in_rec = false; break; }
                rs = re;
                re = p.rec_off[r + 1];
                if (!FILTER) dl = (u32)(rs - p.out_off[r] - gap0);
            }
            const bool live = in_rec && pos < p.total_bases && m < TBX;
            const u64 rem = live ? re - pos : 0;
            const u32 info = (u32)(rem > 127 ? 127 : rem) | ((live && (re - rs) >= (u64)k) ? 128u : 0u);
            if (m < TBX) {
                s_info[m] = (u8)info;
                if (!FILTER) s_dl[m] = dl;
            }
            const u32 c = (m < TBX) ? s_b[m] : 0;
            if (HASH) {
                // position m contributes ror(seed, m) to P and rol(cseed, m) to Q
                const u32 si = (u32)s_t.lut[c] >> 4;
                accP ^= ror64(s_t.seed[si], (u32)m);
                accQ ^= rol64(s_t.cseed[si], (u32)m);
                lp[s] = accP;
                lq[s] = accQ;
            } else {
                const u32 b = (u32)s_t.lut[c] & 15u;
                // first base of a k-mer is most significant: position m goes to bits [2*(15 - m%16)] of word m/16.
                // A thread's PPT consecutive positions touch at most two words: accumulate in registers and
                // issue one LDS atomic per touched word (not one per position)
                if (m < TBX) {
                    const u32 bits = (b & 3u) << (2 * (15 - (m & 15)));
                    if ((m >> 4) == (m0 >> 4)) pk_w0 |= bits; else pk_w1 |= bits;
                    const u32 bb = (b > 3) ? (1u << (m & 31)) : 0u;
                    if ((m >> 5) == (m0 >> 5)) bad_w0 |= bb; else bad_w1 |= bb;
                }
            }
        }
        if (!HASH) {
            if (pk_w0) atomicOr(&s_pk[m0 >> 4], pk_w0);
            if (pk_w1) atomicOr(&s_pk[(m0 >> 4) + 1], pk_w1);
            if (bad_w0) atomicOr(&s_bad[m0 >> 5], bad_w0);
            if (bad_w1) atomicOr(&s_bad[(m0 >> 5) + 1], bad_w1);
        }
        if (HASH) {
            // block-wide exclusive XOR scan of the per-thread totals
            const u64 iP = wave_incl_xor_scan_u64(accP), iQ = wave_incl_xor_scan_u64(accQ);
            if (lane == 63) { s_wtot[wave] = iP; s_wtot[NWV + wave] = iQ; }
            __syncthreads();
            u64 eP = iP ^ accP, eQ = iQ ^ accQ;
#pragma unroll
            for (int w = 0; w < NWV; w++)
                if (w < wave) { eP ^= s_wtot[w]; eQ ^= s_wtot[NWV + w]; }
            if (tid == 0) { s_P[0] = 0; s_Q[0] = 0; }
#pragma unroll
            for (int s = 0; s < PPT; s++)
                if (m0 + s < TBX) {
                    s_P[m0 + s + 1] = eP ^ lp[s];
                    s_Q[m0 + s + 1] = eQ ^ lq[s];
                }
        }
    }
    __syncthreads();

    // ---- phase 2: striped over windows i = tid + j*NT ---------------------------------------------------
    const u64 out0 = FILTER ? 0 : P0 - s_r[2];
    u64 hv[WPT];
    u32 keep = 0, illegal = 0;
    const u64 kmask = (k == 64) ? ~0ull : ((1ull << k) - 1);
#pragma unroll
    for (int j = 0; j < WPT; j++) {
        const int i = tid + j * NT;
        const u32 info = s_info[i];
        const bool long_enough = (info & 128u) != 0;
        const bool fits = (int)(info & 127u) >= k;
        bool emit = long_enough && fits;
        // the common case is computed unconditionally (no exec-mask branches around it)
        u64 f, rv;
        bool bad = false;
        if (HASH) {
            f = rol64(s_P[i + k] ^ s_P[i], (u32)(i + k - 1));
            rv = ror64(s_Q[i + k] ^ s_Q[i], (u32)i);
        } else {
            // 2k-bit field starting at packed bit offset 2i (big-endian within 32-bit words)
            const int w0 = i >> 4, sh = 2 * (i & 15);
            const u64 hi = ((u64)s_pk[w0] << 32) | s_pk[w0 + 1];
            const u64 lo = ((u64)s_pk[w0 + 2] << 32);
            const u64 x = (hi << sh) | ((lo >> 1) >> (63 - sh));  // 64 bits = 32 bases from position i (sh may be 0)
            f = x >> (64 - 2 * k);
            rv = revcomp2(f, k);
            // any illegal base among positions [i, i+k)?
            const int b0 = i >> 5, bs = i & 31;
            const u64 mlo = ((u64)s_bad[b0 + 1] << 32) | s_bad[b0];
            const u64 mhi = s_bad[b0 + 2];
            const u64 mbits = (mlo >> bs) | ((mhi << 1) << (63 - bs));
            bad = (mbits & kmask) != 0;
        }
        if (long_enough && !fits && p.circular) {  // wrapped window: recompute from the record (k-1 per record)
            const u64 pos = P0 + (u64)i;
            const u64 hi_r = (s_r[1] + 1 < p.n_rec + 1) ? s_r[1] + 1 : p.n_rec + 1;
            const u64 r = upper_bound_u64(p.rec_off, s_r[0], hi_r, pos) - 1;
            const u64 rs = p.rec_off[r], len = p.rec_off[r + 1] - rs;
            const u64 o = pos - rs;
            f = 0; rv = 0; bad = false;
            for (int q = 0; q < k; q++) {
                u64 z = o + (u64)q;
                if (z >= len) z -= len;
                const u32 cc = p.bases[rs + z];
                if (HASH) {
                    const u32 si = (u32)s_t.lut[cc] >> 4;
                    f ^= rol64(s_t.seed[si], (u32)(k - 1 - q));
                    rv ^= rol64(s_t.cseed[si], (u32)q);
                } else {
                    const u32 bb = (u32)s_t.lut[cc] & 15u;
                    if (bb > 3) bad = true;
                    f = (f << 2) | (u64)(bb & 3);
                    rv |= (u64)(3 - (bb & 3)) << (2 * q);
                }
            }
            emit = true;
        }
        if (bad && emit) illegal = 1;
        const u64 v = (p.canonical && rv < f) ? rv : f;
        if (FILTER) {
            hv[j] = v;
            if (emit && v <= p.max_hash) keep |= 1u << j;
        } else if (emit) {
            const u64 oi = out0 + (u64)i - (u64)s_dl[i];
            if (oi < p.out_cap) p.out[oi] = v;
        }
    }
    if (illegal) atomicOr((unsigned long long *)&p.result[1], 1ull);

    if (FILTER) {
        // survivors in window order: (round j, wave, lane)
        u32 before[WPT];
#pragma unroll
        for (int j = 0; j < WPT; j++) {
            const u64 m = __ballot((keep >> j) & 1u);
            before[j] = (u32)__popcll(m & ((1ull << lane) - 1));
            if (lane == 0) s_cnt[j * NWV + wave] = (u32)__popcll(m);
        }
        __syncthreads();
        if (tid < 64) {
            constexpr int NG = WPT * NWV;
            const u32 c = lane < NG ? s_cnt[lane] : 0;
            const u32 incl = wave_incl_scan_u32(c);
            if (lane < NG) s_cnt[lane] = incl - c;
            if (lane == 63) s_cnt[NG] = incl;
        }
        __syncthreads();
        const u32 tile_total = s_cnt[WPT * NWV];
        if (tid < 64) {
            bool timed_out = false;
            if (lane == 0) lb_publish(p.status, tile, (u64)tile_total);
            const u64 base = lb_resolve(p.status, tile, (u64)tile_total, lane, TICKET ? nullptr : &timed_out);
            if (tid == 0) s_misc[1] = base;
            if (timed_out && lane == 0) atomicOr((unsigned long long *)&p.result[1], 2ull);
        }
        __syncthreads();
        const u64 base = s_misc[1];
#pragma unroll
        for (int j = 0; j < WPT; j++)
            if ((keep >> j) & 1u) {
                const u64 pos = base + s_cnt[j * NWV + wave] + before[j];
                if (pos < p.out_cap) p.out[pos] = hv[j];
            }
        if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
    }
}

// ---- Scaled-MinHash sketch of reads: per-lane ROLLING ntHash over strips ------------------------------------
// The prefix-XOR kernel above spends ~160 lane-operations per window (per-position prefix words, a block
// scan, two 64-bit variable rotates per window) although with `--scale 1000` only one window in a thousand
// is ever written.  Here every lane rolls the hash along its own strip of L window starts:
//     fwd' = rol(fwd, 1) ^ rol(seed[out], k) ^ seed[in]
//     rev' = ror(rev, 1) ^ ror(seed[comp(out)], 1) ^ rol(seed[comp(in)], k - 1)          
// The hash of a window is a function of its k bytes only, so the rolling runs straight across record
// boundaries of the concatenated base array; whether a window lies inside one record is decided for the few
// values that pass `<= max_hash` (a binary search per CANDIDATE, not per position).  Per step: two byte
// extracts, two 16-byte LDS table reads (tables indexed by the raw byte: no base translation at all), four
// 32-bit funnel shifts, four 3-input XORs, two 32-bit compares: ~16 lane-operations per base.  Bases are read
// straight from global memory, 64 bytes per lane at a time (each lane walks its own strip; a lane's 64-byte
// piece is one cache line, fetched once).  A strip starts cold at its first window start, i.e. it processes
// L + k - 1 bases for L windows.
// Candidates are appended to an LDS list (owner lane, sequence number in the lane), ordered by (lane,
// sequence) = window order with one block scan, validated against the record table, compacted with
// ballots and placed with the library's look-back: the output keeps window order, as the reference's
// iterator does (count.go:361,373-375).  More candidates than the list holds (a tiny --scale, pathological
// repeats) -> flag, and the caller takes the prefix-XOR kernel.
constexpr int ST_NT = 256;
#ifndef ST_CAP_N
#define ST_CAP_N 1024
#endif
constexpr int ST_CAP = ST_CAP_N;         // candidate list entries per tile
constexpr int ST_RND = ST_CAP / ST_NT;   // compaction rounds
constexpr int ST_NWV = ST_NT / 64;

struct StripArgs {
    const u8 *bases;
    const u64 *rec_off;
    u64 n_rec;
    u64 total_bases;
    int k;
    int canonical;
    int L;  // window starts per lane, multiple of 64
    u64 max_hash;
    u64 *out;
    u64 out_cap;
    u64 *status;
    u32 *ticket;
    u64 *result;  // [0] total, [1] flags: bit1 = look-back watchdog, bit2 = candidate list overflow
    u64 ntiles;
    const u64 *tile_rec;
};

struct __attribute__((packed, aligned(4))) U4a4 {  // 16 bytes at 4-byte alignment
    u32 x, y, z, w;
};

// n dwords of the base array starting at byte offset `off` (may be negative or run past the end: such
// bytes read as 0).  Fast path: plain vector loads.
template <int N>
__device__ __forceinline__ void strip_load(const u8 *bases, long long off, u64 total, bool active, u32 (&w)[N]) {
    const bool fast = active && off >= 0 && (u64)off + 4ull * N <= total;
    if (fast) {
        const u8 *src = bases + off;
#pragma unroll
        for (int g = 0; g + 4 <= N; g += 4) {
            const U4a4 q = *reinterpret_cast<const U4a4 *>(src + 4 * g);
            w[g] = q.x; w[g + 1] = q.y; w[g + 2] = q.z; w[g + 3] = q.w;
        }
#pragma unroll
        for (int g = N & ~3; g < N; g++) w[g] = *reinterpret_cast<const u32 *>(src + 4 * g);
    } else {
        // byte by byte; the valid byte indices [ilo, ihi) as two ints so that nothing 64-bit is kept per byte
        const long long lo = -off, hi = (long long)total - off;
        const int ilo = lo < 0 ? 0 : (lo > 4 * N ? 4 * N : (int)lo);
        const int ihi = !active || hi < 0 ? 0 : (hi > 4 * N ? 4 * N : (int)hi);
        const u8 *src = bases + off;
#pragma unroll
        for (int g = 0; g < N; g++) {
            u32 x = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4 * g + q >= ilo && 4 * g + q < ihi) x |= (u32)src[4 * g + q] << (8 * q);
            w[g] = x;
        }
    }
}

template <bool TICKET>
__global__ __launch_bounds__(ST_NT) void nthash_strip_kernel(StripArgs p) {
    __shared__ __attribute__((aligned(16))) uint4 s_tin[256];   // [byte] = { seed, rol(cseed, k-1) }
    __shared__ __attribute__((aligned(16))) uint4 s_tout[256];  // [byte] = { rol(seed, k), ror(cseed, 1) }
    __shared__ u64 s_ch[ST_CAP];   // candidates in arrival order
    __shared__ u32 s_ci[ST_CAP];   // owner lane << 24 | sequence number << 11 | window index in the strip
    __shared__ unsigned short s_ord[ST_CAP];  // window order -> arrival order (an index, so that the list is not copied)
    __shared__ u32 s_base[ST_NT];
    __shared__ u32 s_scan[ST_NWV + 1];
    __shared__ u32 s_cnt[ST_RND * ST_NWV + 1];
    __shared__ u32 s_n;
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int k = p.k, L = p.L;
    {
        const u32 si = (u32)g_byte_table.v[tid] >> 4;  // 0..3 = A C G T/U, 4 = anything else (zero seed)
        const u64 f = si == 0 ? SEED_A : si == 1 ? SEED_C : si == 2 ? SEED_G : si == 3 ? SEED_T : 0ull;
        const u64 cs = si == 0 ? SEED_T : si == 1 ? SEED_G : si == 2 ? SEED_C : si == 3 ? SEED_A : 0ull;
        const u64 a = f, b = rol64(cs, (u32)(k - 1)), c2 = rol64(f, (u32)k), d = ror64(cs, 1);
        s_tin[tid] = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
        s_tout[tid] = make_uint4((u32)c2, (u32)(c2 >> 32), (u32)d, (u32)(d >> 32));
        if (tid == 0) s_n = 0;
    }
    u64 tile = blockIdx.x;
    if (TICKET) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    }
    __syncthreads();
    const u64 P0 = tile * (u64)ST_NT * (u64)L;
    const u64 s0 = P0 + (u64)tid * (u64)L;  // first window start of this lane
    const bool active = s0 < p.total_bases;
    const u32 mh_hi = (u32)(p.max_hash >> 32);
    const bool canon = p.canonical != 0;
    const u32 rmask = canon ? 0u : 0xFFFFFFFFu;  // forward hashes only: the reverse strand never passes the coarse test
    u32 flo = 0, fhi = 0, rlo = 0, rhi = 0;  // fwd / rev hash of the k bases ending at the current position
    u32 myseq = 0;
    const int nsteps = L + k - 1;
    const int kp = (int)((0u - (u32)k) & 3u);  // byte phase of the `out` stream inside its dwords
    for (int c0 = 0; c0 < nsteps; c0 += 64) {
        u32 iw[16], ow[17];
        strip_load<16>(p.bases, (long long)(s0 + (u64)c0), p.total_bases, active, iw);
        // out bytes of this chunk = base[s0 + c0 - k ...]; the dword-aligned window that holds them
        const long long ooff = (long long)(s0 + (u64)c0) - (long long)k;
        strip_load<17>(p.bases, ooff - (long long)kp, p.total_bases, active, ow);
        if (c0 == 0) {
            // a strip starts cold: the first k steps have no outgoing base (byte 0 has a zero table entry)
#pragma unroll
            for (int g = 0; g < 17; g++) {
                const int b0 = 4 * g - kp;        // out-stream step of the dword's first byte
                const int nz = k - b0;            // bytes of this dword that belong to steps < k
                const u32 m = nz <= 0 ? 0xFFFFFFFFu : (nz >= 4 ? 0u : (0xFFFFFFFFu << (8 * nz)));
                ow[g] &= m;
            }
        }
#pragma unroll
        for (int g = 0; g < 16; g++) {
            // the four out bytes of steps 4g .. 4g+3
            const u32 o4 = __builtin_amdgcn_alignbyte(ow[g + 1], ow[g], (u32)kp);
            const u32 i4 = iw[g];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = c0 + 4 * g + q;  // step: the base at s0 + i enters
                const uint4 ein = s_tin[(i4 >> (8 * q)) & 0xFFu];
                const uint4 eout = s_tout[(o4 >> (8 * q)) & 0xFFu];
                // fwd = rol(fwd, 1) ^ seed[in] ^ rol(seed[out], k)
                const u32 nflo = xor3(__builtin_amdgcn_alignbit(flo, fhi, 31), ein.x, eout.x);
                const u32 nfhi = xor3(__builtin_amdgcn_alignbit(fhi, flo, 31), ein.y, eout.y);
                // rev = ror(rev, 1) ^ rol(cseed[in], k - 1) ^ ror(cseed[out], 1)
                const u32 nrlo = xor3(__builtin_amdgcn_alignbit(rhi, rlo, 1), ein.z, eout.z);
                const u32 nrhi = xor3(__builtin_amdgcn_alignbit(rlo, rhi, 1), ein.w, eout.w);
                flo = nflo; fhi = nfhi; rlo = nrlo; rhi = nrhi;
                // Coarse test on the high words with ONE vector compare; everything else — also the uniform range
                // check of the step index — only inside the rare hit.  (Written as `hit && i >= k - 1 && i < nsteps`
                // the compiler evaluated the scalar range checks and a short-circuit branch for the second strand on
                // EVERY step: 27 scalar instructions per step against 19 vector ones, and a CU issues one scalar
                // instruction per cycle: the kernel was bound by its scalar unit.)
                const u32 hmin = fhi < (rhi | rmask) ? fhi : (rhi | rmask);
                if (__builtin_expect(hmin <= mh_hi, 0)) {
                    int ii = i;
                    asm volatile("" : "+s"(ii));  // keeps the range check inside the branch
                    if (ii >= k - 1 && ii < nsteps) {
                    const u64 f = ((u64)fhi << 32) | flo, rv = ((u64)rhi << 32) | rlo;
                    const u64 h = (canon && rv < f) ? rv : f;
                    const u32 w = (u32)(i - (k - 1));  // window index inside the strip
                    if (h <= p.max_hash && s0 + (u64)w + (u64)k <= p.total_bases) {
                        const u32 e = atomicAdd(&s_n, 1u);
                        if (e < (u32)ST_CAP) {
                            s_ch[e] = h;
                            s_ci[e] = ((u32)tid << 24) | ((myseq & 0x1FFFu) << 11) | w;
                        }
                        myseq++;
                    }
                    }
                }
            }
        }
    }
    __syncthreads();
    const u32 n_all = s_n;
    const u32 n = n_all < (u32)ST_CAP ? n_all : (u32)ST_CAP;
    if (n_all > (u32)ST_CAP && tid == 0) atomicOr((unsigned long long *)&p.result[1], 4ull);
    // window order = (owner lane, sequence number)
    {
        u32 tot;
        const u32 excl = block_excl_scan_u32<ST_NT>(myseq, s_scan, &tot);
        s_base[tid] = excl;
    }
    __syncthreads();
    for (u32 e = (u32)tid; e < n; e += ST_NT) {
        const u32 ci = s_ci[e];
        const u32 owner = ci >> 24, seq = (ci >> 11) & 0x1FFFu, w = ci & 0x7FFu;
        const u32 slot = s_base[owner] + seq;
        (void)w;
        if (slot < (u32)ST_CAP) s_ord[slot] = (unsigned short)e;
    }
    __syncthreads();
    // is the window inside one record?  (records shorter than k never hold one)
    u64 hv[ST_RND];
    u32 keep = 0;
    {
        const u64 r_lo = p.tile_rec[tile];
        u64 r_hi = p.tile_rec[tile + 1] + 2;
        r_hi = r_hi < p.n_rec + 1 ? r_hi : p.n_rec + 1;
#pragma unroll
        for (int m = 0; m < ST_RND; m++) {
            const u32 slot = (u32)tid + (u32)m * ST_NT;
            hv[m] = 0;
            if (slot < n && n_all <= (u32)ST_CAP) {
                const u32 e = s_ord[slot];
                const u32 ci = s_ci[e];
                const u64 pos = P0 + (u64)(ci >> 24) * (u64)L + (u64)(ci & 0x7FFu);
                const u64 ub = upper_bound_u64(p.rec_off, r_lo, r_hi, pos);  // first record starting behind pos
                if (ub > 0 && ub <= p.n_rec && pos + (u64)k <= p.rec_off[ub]) {
                    keep |= 1u << m;
                    hv[m] = s_ch[e];
                }
            }
        }
    }
    u32 before[ST_RND];
#pragma unroll
    for (int m = 0; m < ST_RND; m++) {
        const u64 b = __ballot((keep >> m) & 1u);
        before[m] = (u32)__popcll(b & ((1ull << lane) - 1));
        if (lane == 0) s_cnt[m * ST_NWV + wave] = (u32)__popcll(b);
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int NG = ST_RND * ST_NWV;
        const u32 c = lane < NG ? s_cnt[lane] : 0;
        const u32 incl = wave_incl_scan_u32(c);
        if (lane < NG) s_cnt[lane] = incl - c;
        if (lane == 63) s_cnt[NG] = incl;
    }
    __syncthreads();
    const u32 tile_total = s_cnt[ST_RND * ST_NWV];
    if (tid < 64) {
        bool timed_out = false;
        if (lane == 0) lb_publish(p.status, tile, (u64)tile_total);
        const u64 base = lb_resolve(p.status, tile, (u64)tile_total, lane, TICKET ? nullptr : &timed_out);
        if (tid == 0) s_misc[1] = base;
        if (timed_out && lane == 0) atomicOr((unsigned long long *)&p.result[1], 2ull);
    }
    __syncthreads();
    const u64 base = s_misc[1];
#pragma unroll
    for (int m = 0; m < ST_RND; m++)
        if ((keep >> m) & 1u) {
            const u64 pos = base + s_cnt[m * ST_NWV + wave] + before[m];
            if (pos < p.out_cap) p.out[pos] = hv[m];
        }
    if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
}

// returns UKM_OK with *done = false when the strip kernel does not apply (or overflowed): the caller then
// runs the general kernel
int run_strip_filter(ukm_ctx *c, const u8 *bases, const u64 *rec_off, u64 n_rec, int k, int canonical, u64 max_hash,
                     u64 *out, u64 out_cap, u64 *n_out, u64 total_bases, bool *done) {
    *done = false;
    // developer / test knob: UKM_NTHASH_STRIP=0 never, =1 always (whatever the scale; overflow still falls back)
    const char *fe = ukm_env(c, "UKM_NTHASH_STRIP");
    const int force = fe ? atoi(fe) : -1;
    if (force == 0) return UKM_OK;
    if (((uintptr_t)bases & 3) != 0) return UKM_OK;
    // expected candidates per tile = NT * L * (max_hash / 2^64), twice that when canonical (min(f, r) <= m iff
    // f <= m or r <= m); keep it below 0.6 of the list
    const double frac = ((double)max_hash + 1.0) / 18446744073709551616.0 * (canonical ? 2.0 : 1.0);
    int L = 1024;
    // enough tiles to fill the chip
    while (L > 256 && total_bases / ((u64)ST_NT * (u64)L) < 2048) L >>= 1;
    // (the count is Poisson: 0.6 x capacity leaves more than ten standard deviations of head room)
    while (L > 64 && (double)ST_NT * L * frac > ST_CAP * 0.6) L >>= 1;
    if (force != 1 && ((double)ST_NT * L * frac > ST_CAP * 0.6 || L < 256)) return UKM_OK;  // small --scale: general kernel
    if (const char *le = ukm_env(c, "UKM_STRIP_L")) L = std::min(1024, std::max(64, atoi(le) / 64 * 64));  // developer knob; s_ci packs the window index into 11 bits
    const u64 tile_pos = (u64)ST_NT * (u64)L;
    const u64 ntiles = (total_bases + tile_pos - 1) / tile_pos;
    if (ntiles > 0x7FFFFFFFull) return UKM_OK;
    u64 *ctl = nullptr, *tile_rec = nullptr;
    const size_t nctl = 8 + lb_status_words(ntiles);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_TRY(ws_alloc_t(c, ntiles + 3, &tile_rec));
    hipLaunchKernelGGL(tile_first_rec_kernel, dim3((unsigned)((ntiles + 3 + 255) / 256)), dim3(256), 0, c->stream, rec_off,
                       n_rec, total_bases, ntiles, tile_pos, tile_rec);
    StripArgs p;
    memset(&p, 0, sizeof(p));
    p.bases = bases; p.rec_off = rec_off; p.n_rec = n_rec; p.total_bases = total_bases; p.k = k;
    p.canonical = canonical; p.L = L; p.max_hash = max_hash; p.out = out; p.out_cap = out_cap;
    p.result = ctl; p.ticket = (u32 *)(ctl + 2); p.status = ctl + 8; p.ntiles = ntiles; p.tile_rec = tile_rec;
    u64 res[2] = {0, 0};
    for (int attempt = c->setop_force_ticket ? 1 : 0; attempt < 2; attempt++) {
        UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
        (void)hipEventRecord(c->ev_k0, c->stream);
        if (attempt == 0) hipLaunchKernelGGL(nthash_strip_kernel<false>, dim3((unsigned)ntiles), dim3(ST_NT), 0, c->stream, p);
        else hipLaunchKernelGGL(nthash_strip_kernel<true>, dim3((unsigned)ntiles), dim3(ST_NT), 0, c->stream, p);
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        UKM_TRY(ukm_read_u64(c, ctl, res, 2));
        if (!(res[1] & 2)) break;
        if (attempt == 1) UKM_FAIL(UKM_ERR_HIP, "ntHash strip kernel: look-back watchdog fired in the ticketed kernel");
        ukm_switch_to_tickets(c, "ntHash strip kernel");
    }
    if (res[1] & 4) return UKM_OK;  // candidate list overflow: general kernel
    *n_out = res[0];
    if (*n_out > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu", (unsigned long long)*n_out,
                 (unsigned long long)out_cap);
    *done = true;
    return UKM_OK;
}

// ---- every window of long records: per-lane rolling 2-bit codes / ntHash over strips ----------------------------
// window_kernel above stages 2048 positions per tile and re-derives every window from per-position prefix
// words: ~45 (codes) to ~160 (ntHash) lane-operations per window, 30 % of the HBM roofline.  For inputs made of
// long records (genomes, contigs: count.go's per-record loop over a chromosome) every lane instead rolls along
// its own strip of L positions:  code' = ((code << 2) | c) & mask,  rc' = (rc >> 2) | ((3 - c) << (2k - 2)),
// or the ntHash recurrence of nthash_strip_kernel.  A lane owns the windows that END inside its strip, so
// step i of the unrolled loop always fills slot i & 15 of the lane's LDS row whatever k is.  Every 16 steps
// the wave turns its 64 rows x 16 values around: eight lanes write one row's 128 bytes with 16-byte stores.
// Neighbouring lanes work 8L bytes apart, so what matters is that every row is one whole, ALIGNED cache line
// of the output (measured: rows that straddle lines cost 2.3 x; on 150-bp reads, k = 31: rows on 128 / 64 / 32 /
// 16 / 8-byte boundaries 0.206 / 0.230 / 0.30 / 0.33 / 0.35 ms per 1e8 windows): every lane starts its strip `sh`
// (< 16) positions early so that the output index of its row starts is a multiple of 16 in the record it starts
// in; the tile's first lane skips what belongs to the previous tile, every wave runs 16 steps longer.
// All record logic (is the window inside one record, where does it go) happens once per row, not per window:
// a row that lies inside one record gets its output index published; a row that touches a record or strip
// boundary (rare for long records) is written by its owner lane value by value.
// Not for circular records; more than SW_REC records under one tile -> flag, the caller runs window_kernel.
constexpr int SW_NT = 256;
constexpr int SW_NWV = SW_NT / 64;
constexpr int SW_REC = 254;   // records one tile may touch
constexpr int SW_B = 16;      // values per row = one 128-byte line
constexpr int SW_ROW = 17;    // u64 per LDS row: 16 values + 1 pad (bank spread)
#ifndef SW_ABL
#define SW_ABL 0              // developer ablations: 1 = no row stores
#endif

struct SwArgs {
    const u8 *bases;
    const u64 *rec_off;   // [n_rec + 1]
    const u64 *out_off;   // [n_rec + 1] exclusive scan of the per-record window counts
    u64 n_rec;
    u64 total_bases;
    int k;
    int canonical;
    int L;                // positions per lane, multiple of 64
    u64 *out;
    u64 *result;          // [1] flags: bit0 illegal base in an emitted window, bit2 record table overflow
    const u64 *tile_rec;  // [ntiles + 3]
    // ukm_count (round 6): the sort that follows wants the 256-bin histogram of digit (value >> fshift) & 255 of everything this
    // launch writes -- counted here (one LDS atomic per value, 256 global atomics per workgroup) it saves the sort's own
    // pre-pass over the values; null: not wanted
    u64 *fhist;
    int fshift;
};

template <bool HASH>
__global__ __launch_bounds__(SW_NT) __attribute__((amdgpu_waves_per_eu(HASH ? 3 : 4, HASH ? 3 : 4))) void stripwin_kernel(SwArgs p) {
    __shared__ __attribute__((aligned(16))) uint4 s_tin[HASH ? 256 : 1];
    __shared__ __attribute__((aligned(16))) uint4 s_tout[HASH ? 256 : 1];
    __shared__ u8 s_lut[HASH ? 1 : 256];
    __shared__ u64 s_ro[SW_REC + 2];
    __shared__ u32 s_gap[SW_REC + 2];  // gap of a record minus the gap of the tile's first record (< tile positions + k per record)
    __shared__ u64 s_row[SW_NWV][64 * SW_ROW];
    __shared__ u64 s_desc[SW_NWV][64];
    __shared__ u32 s_fh[128];  // fhist: two 16-bit counters per word (a tile writes at most SW_NT x L <= 65,535 values: the host checks)
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int k = p.k, L = p.L;
    if (p.fhist && tid < 128) s_fh[tid] = 0;
    if constexpr (HASH) {
        const u32 si = (u32)g_byte_table.v[tid] >> 4;
        const u64 f = si == 0 ? SEED_A : si == 1 ? SEED_C : si == 2 ? SEED_G : si == 3 ? SEED_T : 0ull;
        const u64 cs = si == 0 ? SEED_T : si == 1 ? SEED_G : si == 2 ? SEED_C : si == 3 ? SEED_A : 0ull;
        const u64 a = f, b = rol64(cs, (u32)(k - 1)), c2 = rol64(f, (u32)k), d = ror64(cs, 1);
        s_tin[tid] = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
        s_tout[tid] = make_uint4((u32)c2, (u32)(c2 >> 32), (u32)d, (u32)(d >> 32));
    } else {
        s_lut[tid] = g_byte_table.v[tid] & 0xF;
    }
    const u64 tile = blockIdx.x;
    const u64 TS = (u64)SW_NT * (u64)L;
    const u64 P0 = tile * TS;
    const u64 r0 = p.tile_rec[tile];
    const u64 r1 = p.tile_rec[tile + 1];
    const u64 nr = r1 - r0 + 1;  // records r0 .. r1 may hold positions of the tile
    if (nr > (u64)SW_REC) {
        if (tid == 0) atomicOr((unsigned long long *)&p.result[1], 4ull);
        return;
    }
    const u64 gap_first = p.rec_off[r0] - p.out_off[r0];  // (uniform: scalar loads)
    for (u64 j = (u64)tid; j <= nr; j += SW_NT) {
        const u64 ro = p.rec_off[r0 + j];
        s_ro[j] = ro;
        s_gap[j] = (u32)((ro - p.out_off[r0 + j]) - gap_first);
    }
    __syncthreads();
    // output index of the window that ends at e (inside record r) = e + 1 - k - gap[r].  Every lane starts its strip
    // `sh` (< 16) positions in front of its nominal start b = P0 + tid * L so that this index is a multiple of 16 at
    // each of its row starts, FOR THE RECORD THAT HOLDS b: the lane's rows are then whole output lines until it walks
    // into the next record (reads: 8 % of the rows of 64-position strips; one shift per tile, the first version, left
    // every record but the tile's first one at the alignment its gap happens to have).  The boundaries between lanes
    // move with the shifts (s_bnd); the tile's own boundaries stay where they are.
    const u64 b_nom = P0 + (u64)tid * (u64)L;
    u32 sh = 0, jb = 0;
    bool active = b_nom < p.total_bases;
    if (active) {
        active = false;
        u32 lo = 0, hi = (u32)nr + 1;  // first index with s_ro > b_nom (records of length 0 are skipped)
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (s_ro[mid] <= b_nom) lo = mid + 1; else hi = mid;
        }
        if (lo >= 1 && lo <= (u32)nr) {
            active = true;
            jb = lo - 1;
            sh = (u32)(b_nom + 1 - (u64)k - (gap_first + (u64)s_gap[jb])) & (u32)(SW_B - 1);
        }
    }
    u32 *s_bnd = reinterpret_cast<u32 *>(&s_desc[0][0]);  // first END position of every lane, relative to the tile
    s_bnd[tid] = tid == 0 ? 0u : (u32)tid * (u32)L - sh;  // (tid >= 1: L >= 64 > sh)
    __syncthreads();
    const long long s0 = (long long)b_nom - (long long)sh;                  // first END position of this lane's rows
    const u64 e_lo = P0 + (u64)s_bnd[tid];                                  // the lane emits END positions [e_lo, e_hi)
    const u64 e_hi = P0 + (tid == SW_NT - 1 ? TS : (u64)s_bnd[tid + 1]);
    __syncthreads();  // (s_bnd lives in the descriptor words)
    u32 j = 0;
    u64 rec_start = ~0ull, rec_end = 0, gap = 0;
    bool dead = !active;
    if (active) {
        u32 lo = jb + 1;
        if (e_lo < s_ro[jb]) {  // the shift reached into the previous record
            lo = 0;
            u32 hi = jb + 1;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (s_ro[mid] <= e_lo) lo = mid + 1; else hi = mid;
            }
        }
        if (lo == 0 || lo > (u32)nr) dead = true;
        else { j = lo - 1; rec_start = s_ro[j]; rec_end = s_ro[j + 1]; gap = gap_first + (u64)s_gap[j]; }
        if (dead) { rec_start = ~0ull; rec_end = 0; }
    }
    const bool canon = p.canonical != 0;
    const int WU = k <= 32 ? 32 : 64;     // warm-up steps in front of the strip (>= k)
    const int i_start = 64 - WU;          // step i reads the base at s0 - 64 + i
    // a strip is up to 15 positions longer than L (its own shift, its neighbour's none): the wave runs one more row
    // only if one of its lanes needs it (long records: every lane has the same shift, only the tile's last wave does)
    const bool longer = active && e_hi > (u64)s0 + (u64)L;
    const int nsteps = 64 + L + (__ballot(longer) != 0ull ? SW_B : 0);
    u64 fwd = 0, rc = 0, hist = 0;
    u32 flo = 0, fhi = 0, rlo = 0, rhi = 0;
    const u64 kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const u32 rcs = (u32)(2 * k - 2);
    const u32 ipa = (u32)(s0 - 64) & 3u;                    // byte phase of the in / out streams in their dwords
    const u32 opa = (u32)(s0 - 64 - (long long)k) & 3u;
    u64 *row = &s_row[wave][lane * SW_ROW];
    bool illegal = false;
#define SW_NEXT_RECORD()                                                                  \
    do {                                                                                  \
        if (j + 1 >= (u32)nr) { dead = true; rec_start = ~0ull; rec_end = 0; }            \
        else { j++; rec_start = rec_end; rec_end = s_ro[j + 1]; gap = gap_first + (u64)s_gap[j]; } \
    } while (0)
    for (int c0 = 0; c0 < nsteps; c0 += 64) {
        u32 iw[17];
        u32 ow[HASH ? 17 : 1];
        const long long cpos = s0 + (long long)c0 - 64;
        strip_load<17>(p.bases, cpos - (long long)ipa, p.total_bases, active, iw);
        if constexpr (HASH) {
            u32 t17[17];
            strip_load<17>(p.bases, cpos - (long long)k - (long long)opa, p.total_bases, active, t17);
#pragma unroll
            for (int g = 0; g < 17; g++) ow[g] = t17[g];
        }
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (c0 == 0 && 4 * g < i_start) continue;
            if (c0 + 4 * g >= nsteps) continue;
            u32 o4 = 0;
            if constexpr (HASH) {
                o4 = __builtin_amdgcn_alignbyte(ow[g + 1], ow[g], opa);
                if (c0 == 0) {
                    // cold start: steps < i_start + k have no outgoing base (byte 0 has a zero table entry)
                    const int nz = i_start + k - 4 * g;
                    const u32 m = nz <= 0 ? 0xFFFFFFFFu : (nz >= 4 ? 0u : (0xFFFFFFFFu << (8 * nz)));
                    o4 &= m;
                }
            }
            const u32 i4 = __builtin_amdgcn_alignbyte(iw[g + 1], iw[g], ipa);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u64 v;
                if constexpr (HASH) {
                    const uint4 ein = s_tin[(i4 >> (8 * q)) & 0xFFu];
                    const uint4 eout = s_tout[(o4 >> (8 * q)) & 0xFFu];
                    const u32 nflo = xor3(__builtin_amdgcn_alignbit(flo, fhi, 31), ein.x, eout.x);
                    const u32 nfhi = xor3(__builtin_amdgcn_alignbit(fhi, flo, 31), ein.y, eout.y);
                    const u32 nrlo = xor3(__builtin_amdgcn_alignbit(rhi, rlo, 1), ein.z, eout.z);
                    const u32 nrhi = xor3(__builtin_amdgcn_alignbit(rlo, rhi, 1), ein.w, eout.w);
                    flo = nflo; fhi = nfhi; rlo = nrlo; rhi = nrhi;
                    const u64 f = ((u64)fhi << 32) | flo, rv = ((u64)rhi << 32) | rlo;
                    v = (canon && rv < f) ? rv : f;
                } else {
                    const u32 e = s_lut[(i4 >> (8 * q)) & 0xFFu];
                    const u32 code = e & 3u;
                    hist = (hist << 1) | (u64)(e >> 2);
                    fwd = ((fwd << 2) | (u64)code) & kmask;
                    rc = (rc >> 2) | ((u64)(code ^ 3u) << rcs);
                    v = (canon && rc < fwd) ? rc : fwd;
                }
                row[(4 * g + q) & (SW_B - 1)] = v;
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting a whole chunk's table reads
            if ((g & 3) == 3 && c0 > 0) {
                // ---- one row: END positions e0 .. e0 + 15 ----
                const long long e0s = s0 + (long long)(c0 - 64 + 4 * (g - 3));
                const u64 e0 = (u64)e0s;
                if (k >= 17) {
                    // A window needs k - 1 >= 16 bases in front of its END inside its record, so a row of 16 END
                    // positions holds valid windows of AT MOST ONE record, and they form one run [qlo, qhi) of the
                    // row: rows cut by a record end, a record start or the strip's own limits go through the same
                    // cooperative 16-byte stores as whole rows, with the run packed into the descriptor (round 3; on
                    // 150-bp reads every wave used to spend every row step in a divergent value-by-value loop).
                    u64 dsc = ~0ull;
                    if (!dead && e0s + (SW_B - 1) >= (long long)e_lo && e0s < (long long)e_hi) {
                        const u64 ef = e0s < (long long)e_lo ? e_lo : e0;  // first END of the row this lane may emit
                        while (!dead && ef >= rec_end) SW_NEXT_RECORD();
                        if (!dead) {
                            u64 lo_e = rec_start + (u64)k - 1;
                            lo_e = lo_e < ef ? ef : lo_e;
                            u64 hi_e = e0 + SW_B;
                            hi_e = hi_e < rec_end ? hi_e : rec_end;
                            hi_e = hi_e < e_hi ? hi_e : e_hi;
                            if (lo_e < hi_e) {
                                const u32 qlo = (u32)(lo_e - e0), qhi = (u32)(hi_e - e0);
                                const u64 d = (e0 + 1 - (u64)k - gap) & ((1ull << 55) - 1);  // output index of slot 0 (mod 2^55)
                                dsc = d | ((u64)qlo << 55) | ((u64)qhi << 59);
                                if (!HASH) {
                                    // illegal base inside an emitted window: history bits 16 - qhi .. 14 - qlo + k
                                    const u32 cntb = (u32)k + qhi - qlo - 1;
                                    const u64 hm = (cntb >= 64 ? ~0ull : ((1ull << cntb) - 1)) << (SW_B - qhi);
                                    if ((hist & hm) != 0) illegal = true;
                                }
                            }
                        }
                    }
                    s_desc[wave][lane] = dsc;
                } else {
                bool ok = false;
                if (!dead && e0s >= (long long)e_lo) {
                    while (!dead && e0 >= rec_end) SW_NEXT_RECORD();
                    ok = !dead && e0 + (SW_B - 1) < rec_end && e0 + (SW_B - 1) < e_hi && e0 + 1 >= rec_start + (u64)k;
                }
                s_desc[wave][lane] = ok ? (((e0 + 1 - (u64)k - gap) & ((1ull << 55) - 1)) | ((u64)SW_B << 59)) : ~0ull;
                if (!HASH && ok && (hist & ((1ull << (k + SW_B - 1)) - 1)) != 0) illegal = true;
                if (!ok && !dead && e0s + (SW_B - 1) >= (long long)e_lo && e0s < (long long)e_hi) {
                    // k <= 16: a row can hold windows of several records: the owner writes what is valid, value by value
                    for (int q = 0; q < SW_B; q++) {
                        const long long es = e0s + q;
                        if (es < (long long)e_lo || es >= (long long)e_hi) continue;
                        const u64 e = (u64)es;
                        while (!dead && e >= rec_end) SW_NEXT_RECORD();
                        if (!dead && e + 1 >= rec_start + (u64)k) {
                            p.out[e + 1 - (u64)k - gap] = row[q];
                            if (!HASH && ((hist >> (SW_B - 1 - q)) & ((1ull << k) - 1)) != 0) illegal = true;
                        }
                    }
                }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int strip = m * 8 + (lane >> 3), pr = lane & 7;
                    const u64 dd = s_desc[wave][strip];
                    if (dd != ~0ull && (SW_ABL != 1 || dd == 12345ull)) {
                        // descriptor: output index of slot 0 (55 bits) | first valid slot (4 bits) | end of the run (5 bits)
                        const u32 qlo = (u32)(dd >> 55) & 15u, qhi = (u32)(dd >> 59);
                        const u32 q0 = 2u * (u32)pr, q1 = q0 + 1;
                        const bool w0 = q0 >= qlo && q0 < qhi, w1 = q1 >= qlo && q1 < qhi;
                        const u64 *src = &s_row[wave][strip * SW_ROW + 2 * pr];
                        const u64 v0 = src[0], v1 = src[1];
                        // (each slot's index is reduced on its own: slot 0 of a row that starts in front of its
                        //  record's first window has a "negative" index, and -1 + 1 must come out as 0)
                        const u64 i0 = (dd + q0) & ((1ull << 55) - 1), i1 = (dd + q1) & ((1ull << 55) - 1);
                        if (w0 && w1) {
                            U4a4 st;
                            st.x = (u32)v0; st.y = (u32)(v0 >> 32); st.z = (u32)v1; st.w = (u32)(v1 >> 32);
                            *reinterpret_cast<U4a4 *>(p.out + i0) = st;
                        } else if (w0) {
                            p.out[i0] = v0;
                        } else if (w1) {
                            p.out[i1] = v1;
                        }
                        if (p.fhist) {  // (uniform)
                            const u32 d0 = (u32)(v0 >> p.fshift) & 255u, d1 = (u32)(v1 >> p.fshift) & 255u;
                            if (w0) atomicAdd(&s_fh[d0 >> 1], (d0 & 1u) ? 65536u : 1u);
                            if (w1) atomicAdd(&s_fh[d1 >> 1], (d1 & 1u) ? 65536u : 1u);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#undef SW_NEXT_RECORD
    if (!HASH && illegal) atomicOr((unsigned long long *)&p.result[1], 1ull);
    if (p.fhist) {
        __syncthreads();
        if (tid < 128) {
            const u32 w = s_fh[tid];
            if (w & 0xFFFFu) atomicAdd((unsigned long long *)&p.fhist[2 * tid], (unsigned long long)(w & 0xFFFFu));
            if (w >> 16) atomicAdd((unsigned long long *)&p.fhist[2 * tid + 1], (unsigned long long)(w >> 16));
        }
    }
}

// returns UKM_OK with *done = false when the strip kernel does not apply (short records, tiny input, a tile
// with too many records): the caller then runs window_kernel.  `ctl` is the zeroed control block.
int run_strip_windows(ukm_ctx *c, bool hash, const u8 *bases, const u64 *rec_off, const u64 *out_off, u64 n_rec, int k,
                      int canonical, u64 *out, u64 total_bases, u64 *ctl, bool *done, u64 *fhist = nullptr, int fshift = -1) {
    *done = false;
    const char *fe = ukm_env(c, "UKM_WIN_STRIP");  // developer / test knob: 0 never, 1 whenever it is correct
    const int force = fe ? atoi(fe) : -1;
    if (force == 0) return UKM_OK;
    if (((uintptr_t)bases & 3) != 0 || (!hash && k > 32)) return UKM_OK;
    // small inputs have too few strips to fill the chip (measured cross-over: 8e6 bases for L = 64, 1.6e7 for L = 128;
    // 1.6e7 bases: codes 0.042 ms against 0.060 ms for the general kernel)
    const u64 min_bases = (k <= 32) ? (1ull << 23) : (1ull << 24);
    // Records: the kernel also wins on short ones — 1e8 bases of 150-bp reads: codes 0.20 ms against 0.43 ms for the
    // general kernel, ntHash k = 51 0.25 against 0.51 ms (0.29 / 0.32 ms with value-by-value boundary rows, 0.264 /
    // 0.285 ms with one strip shift per tile, round 3) — as long as a tile's records fit its table (254 per 256 x L
    // positions, else the call falls back after one wasted launch): average length >= 80 bases for L = 64, >= 140
    // for L = 128.
    const u64 min_avg = (k <= 32) ? 80 : 140;
    if (force != 1 && (total_bases < min_bases || n_rec * min_avg > total_bases)) return UKM_OK;
    // several rounds of workgroups per CU matter more than the k - 1 warm-up steps per strip (measured at 1e8
    // bases, codes: L = 64 / 128 / 256 / 512 -> 0.178 / 0.185 / 0.21 / 0.22 ms; ntHash k = 51: 0.242 / 0.230 / 0.244)
    // (1e9 bases: codes 1.71 / 1.84 / 1.94 ms, ntHash k = 51 2.23 / 1.89 / 1.96 ms: short strips keep a wave's 64 output
    // rows close together in memory, which is worth more than the shorter warm-up of long ones)
    int L = (k <= 32) ? 64 : 128;  // twice the warm-up
    if (const char *le = ukm_env(c, "UKM_WIN_STRIP_L")) L = std::max(64, atoi(le) / 64 * 64);
    const u64 tile_pos = (u64)SW_NT * (u64)L;
    const u64 ntiles = (total_bases + tile_pos - 1) / tile_pos;
    if (ntiles > 0x7FFFFFFFull) return UKM_OK;
    u64 *tile_rec = nullptr;
    UKM_TRY(ws_alloc_t(c, ntiles + 3, &tile_rec));
    hipLaunchKernelGGL(tile_first_rec_kernel, dim3((unsigned)((ntiles + 3 + 255) / 256)), dim3(256), 0, c->stream, rec_off,
                       n_rec, total_bases, ntiles, tile_pos, tile_rec);
    SwArgs p;
    memset(&p, 0, sizeof(p));
    p.bases = bases; p.rec_off = rec_off; p.out_off = out_off; p.n_rec = n_rec; p.total_bases = total_bases;
    p.k = k; p.canonical = canonical; p.L = L; p.out = out; p.result = ctl; p.tile_rec = tile_rec;
    // (k <= 16 writes the rows a record cuts value by value, without the count; 16-bit counters: a tile of <= 65,535 positions)
    if (fhist && fshift >= 0 && k >= 17 && tile_pos <= 65535) { p.fhist = fhist; p.fshift = fshift; }
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (hash) hipLaunchKernelGGL(stripwin_kernel<true>, dim3((unsigned)ntiles), dim3(SW_NT), 0, c->stream, p);
    else hipLaunchKernelGGL(stripwin_kernel<false>, dim3((unsigned)ntiles), dim3(SW_NT), 0, c->stream, p);
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    u64 res[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, res, 2));
    if (res[1] & 4) {  // a tile with more records than the table holds: general kernel
        UKM_HIP(hipMemsetAsync(ctl, 0, 2 * sizeof(u64), c->stream));
        return UKM_OK;
    }
    if (res[1] & 1) UKM_FAIL(UKM_ERR_ILLEGAL_BASE, "illegal base in sequence (kmers.ErrIllegalBase)");
    *done = true;
    return UKM_OK;
}

// fused (may be null; ukm_count): fused->hist = 256 zeroed device words, fused->key_bits = the width the values will be sorted
// by.  On return fused->shift >= 0 says the histogram of digit (value >> shift) & 255 of ALL values written is in fused->hist
// (the strip kernel ran with it); -1: nobody counted, the sort runs its own pre-pass.
struct FusedHist { u64 *hist; int key_bits; int shift; };

int run_windows(ukm_ctx *c, bool hash, const u8 *bases, const u64 *rec_off, u64 n_rec, int k,
                int canonical, int circular, u64 max_hash, u64 *out, u64 out_cap, u64 *n_out,
                u64 total_bases, const u64 **win_off = nullptr, FusedHist *fused = nullptr) {
    *n_out = 0;
    if (fused) fused->shift = -1;
    if (win_off) *win_off = nullptr;
    if (n_rec == 0 || total_bases == 0) return UKM_OK;
    if (hash && max_hash != 0 && !circular && !win_off) {
        // Scaled-MinHash sketch: the rolling strip kernel when the filter is selective enough
        bool done = false;
        UKM_TRY(run_strip_filter(c, bases, rec_off, n_rec, k, canonical, max_hash, out, out_cap, n_out, total_bases, &done));
        if (done) return UKM_OK;
        *n_out = 0;
    }
    // per-record window counts -> exclusive scan (n_rec + 1 entries: off[n_rec] = total)
    u64 *cnt = nullptr, *off = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, n_rec + 1, &cnt));
    UKM_TRY(ws_alloc_t(c, n_rec + 1, &off));
    if (win_off) *win_off = off;
    hipLaunchKernelGGL(window_count_kernel, dim3((unsigned)((n_rec + 1 + 255) / 256)), dim3(256), 0, c->stream,
                       rec_off, n_rec, k, circular, cnt);
    const u64 ntiles = (total_bases + WT - 1) / WT;
    const bool filter = hash && max_hash != 0;
    const size_t nctl = 8 + (filter ? lb_status_words(ntiles) : 0);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
    UKM_TRY(ukm_dev_exclusive_scan_u64(c, cnt, off, n_rec + 1, ctl + 3));
    u64 total_windows = 0;
    UKM_TRY(ukm_read_u64(c, ctl + 3, &total_windows));
    if (total_windows == 0) return UKM_OK;
    if (!filter && total_windows > out_cap) {
        *n_out = total_windows;
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu",
                 (unsigned long long)total_windows, (unsigned long long)out_cap);
    }
    if (!circular && !filter) {
        // long records: the rolling strip kernel
        bool done = false;
        const int fsh = (fused && fused->hist) ? ukm_sort_first_shift(c, total_windows, fused->key_bits) : -1;
        UKM_TRY(run_strip_windows(c, hash, bases, rec_off, off, n_rec, k, canonical, out, total_bases, ctl, &done, fsh >= 0 ? fused->hist : nullptr, fsh));
        if (done) {
            *n_out = total_windows;
            if (fused && fsh >= 0 && k >= 17) fused->shift = fsh;  // (the conditions under which run_strip_windows handed the histogram on)
            return UKM_OK;
        }
        if (fused && fsh >= 0) UKM_HIP(hipMemsetAsync(fused->hist, 0, 256 * sizeof(u64), c->stream));  // (a partial count of the launch that gave up)
    }
    WinArgs p;
    memset(&p, 0, sizeof(p));
    p.bases = bases; p.rec_off = rec_off; p.out_off = off; p.n_rec = n_rec;
    p.total_bases = total_bases; p.k = k; p.canonical = canonical; p.circular = circular;
    p.max_hash = max_hash; p.out = out; p.out_cap = out_cap;
    p.result = ctl; p.ticket = (u32 *)(ctl + 2); p.status = ctl + 8; p.ntiles = ntiles;
    u64 *tile_rec = nullptr;
    UKM_TRY(ws_alloc_t(c, ntiles + 3, &tile_rec));
    hipLaunchKernelGGL(tile_first_rec_kernel, dim3((unsigned)((ntiles + 3 + 255) / 256)), dim3(256), 0, c->stream,
                       rec_off, n_rec, total_bases, ntiles, (u64)WT, tile_rec);
    p.tile_rec = tile_rec;
    u64 res[2] = {0, 0};
    for (int attempt = (filter && c->setop_force_ticket) ? 1 : 0; attempt < 2; attempt++) {
        if (attempt == 1) UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));  // second try: ticketed
        (void)hipEventRecord(c->ev_k0, c->stream);
        if (!hash) hipLaunchKernelGGL((window_kernel<false, false>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        else if (!filter) hipLaunchKernelGGL((window_kernel<true, false>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        else if (attempt == 0) hipLaunchKernelGGL((window_kernel<true, true, false>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        else hipLaunchKernelGGL((window_kernel<true, true, true>), dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        UKM_TRY(ukm_read_u64(c, ctl, res, 2));
        if (!(res[1] & 2)) break;  // no look-back watchdog
        if (attempt == 1) UKM_FAIL(UKM_ERR_HIP, "window kernel: look-back watchdog fired in the ticketed kernel");
        ukm_switch_to_tickets(c, "ntHash filter kernel");  // this device does not dispatch workgroups in order
    }
    if (res[1] & 1) UKM_FAIL(UKM_ERR_ILLEGAL_BASE, "illegal base in sequence (kmers.ErrIllegalBase)");
    *n_out = filter ? res[0] : total_windows;
    if (*n_out > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu",
                 (unsigned long long)*n_out, (unsigned long long)out_cap);
    return UKM_OK;
}


// ---- minimizer sketch (bio/sketches MinimizerSketch; count.go:316,357) ------------------
// Input: the canonical ntHash of every window (h, per-record offsets off[n_rec+1]).  Group g of a
// record = windows [g, g+w); its minimizer is the LEFTMOST minimum; it is emitted when the arg-min
// position differs from the previous group's.  With m = min h[g .. g+w-2] (leftmost):
//   arg(g-1) == g-1        <=>  h[g-1] <= m
//   arg(g)   == g+w-1      <=>  h[g+w-1] <  m
// and otherwise arg(g-1) == arg(g), so  emit(g) = g==0 || h[g-1] <= m || h[g+w-1] < m  -- every
// group is decided independently of the others (no serial scan over the record).
#ifndef MIN_MT
#define MIN_MT 2048  /* 1024 / 2048 / 4096: 1.24 / 1.01 / 1.19 ms per 1e8 windows incl. the ntHash pass (w = 15) */
#endif
constexpr int MT = MIN_MT;      // window indices per tile
constexpr int MPT = MT / NT;    // per thread, striped
constexpr int MW_MAX = 1024;    // largest supported w (LDS halo)

struct MinArgs {
    const u64 *h;
    const u64 *off;  // [n_rec + 1]
    u64 n_rec;
    u64 total;
    int w;
    u64 max_hash;
    u64 *out;
    u64 *out_pos;  // may be NULL
    u64 out_cap;
    u64 *status;
    u32 *ticket;
    u64 *result;
    u64 ntiles;
    const u64 *tile_rec;
};

// The leftmost minimum of h[g .. g+w-2] for every group comes from a sparse table built in place in LDS:
// level t holds the (leftmost) minimum of the 2^t hashes starting at every index; log2(w-1) doubling steps
// (operands to registers | barrier | store | barrier), then two overlapping table entries per group.  The first
// version walked the w - 2 hashes per group: O(w) LDS reads per window (w = 15: 1.0 ms per 1e8 windows, w = 200:
// 3.0 ms) and a binary search in the global record table per window.
constexpr int M_EPT = (MT + MW_MAX + 1 + NT - 1) / NT;  // table entries per thread
constexpr int M_REC = 256;                              // records of a tile kept in LDS

template <bool TICKET>
__global__ __launch_bounds__(NT) void minimizer_kernel(MinArgs p) {
    __shared__ u64 s_h[MT + MW_MAX + 1];            // s_h[i] = h[J0 - 1 + i]; becomes the sparse table's top level
    __shared__ unsigned short s_ix[MT + MW_MAX + 1];  // arg-min (index into s_h) of every table entry
    __shared__ u64 s_off[M_REC + 2];
    __shared__ u32 s_cnt[MPT * NWV + 1];
    __shared__ u64 s_r[2];
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    u64 tile = blockIdx.x;  // see window_kernel: a ticket counter serialises every workgroup on one address
    if (TICKET) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    }
    const u64 J0 = tile * (u64)MT;
    const int w = p.w;
    const int E = MT + w + 1;  // entries in use
    for (int i = tid; i < E; i += NT) {
        const u64 g = J0 + (u64)i;
        s_h[i] = (g >= 1 && g - 1 < p.total) ? p.h[g - 1] : ~0ull;
        s_ix[i] = (unsigned short)i;
    }
    if (tid == 0) {
        s_r[0] = p.tile_rec[tile];
        s_r[1] = p.tile_rec[tile + 2] + 1;
    }
    __syncthreads();
    const u64 r_lo = s_r[0];
    const u64 r_hi = (s_r[1] + 1 < p.n_rec + 1) ? s_r[1] + 1 : p.n_rec + 1;
    // the tile's slice of the record table: off[r_lo .. r_hi]
    // (the search looks at entries r_lo .. r_hi - 1; entry r_hi is read as the end of the last record when it exists)
    const bool rec_in_lds = r_hi - r_lo <= (u64)M_REC;
    if (rec_in_lds) {
        const u64 top_e = r_hi < p.n_rec ? r_hi : p.n_rec;
        for (u64 q = (u64)tid; r_lo + q <= top_e; q += NT) s_off[q] = p.off[r_lo + q];
    }
    // per group: its record, validity, the two hashes outside the inner window (the table overwrites s_h)
    u64 prevh[MPT], lasth[MPT], gpos[MPT];
    u32 valid = 0;
#pragma unroll
    for (int jj = 0; jj < MPT; jj++) {
        const int i = tid + jj * NT;
        prevh[jj] = s_h[i];
        lasth[jj] = s_h[i + w];
        gpos[jj] = 0;
    }
    __syncthreads();  // s_off complete; s_h read
#pragma unroll
    for (int jj = 0; jj < MPT; jj++) {
        const int i = tid + jj * NT;
        const u64 j = J0 + (u64)i;
        if (j >= p.total) continue;
        u64 rs, nwin;
        if (rec_in_lds) {
            u32 lo = 0, hi = (u32)(r_hi - r_lo);  // first entry > j among off[r_lo .. r_hi)
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (s_off[mid] <= j) lo = mid + 1; else hi = mid;
            }
            if (lo == 0 || r_lo + lo > p.n_rec) continue;
            rs = s_off[lo - 1];
            nwin = s_off[lo] - rs;
        } else {
            const u64 ub = upper_bound_u64(p.off, r_lo, r_hi, j);
            if (ub == 0 || ub > p.n_rec) continue;
            rs = p.off[ub - 1];
            nwin = p.off[ub] - rs;
        }
        const u64 g = j - rs;
        if (g + (u64)w > nwin) continue;
        valid |= 1u << jj;
        gpos[jj] = g;
    }
    // ---- sparse table over s_h[1 ..]: after level t, entry i = leftmost minimum of s_h[i .. i + 2^t) ----
    const int W1 = w - 1;  // inner window h[g .. g+w-2]
    int top = 0;
    while ((2 << top) <= W1) top++;  // 2^top <= W1 < 2^(top+1)   (W1 >= 1 here; w == 1 skips the table)
    if (W1 >= 1) {
        for (int t = 0; t < top; t++) {
            const int d = 1 << t;
            u64 nv[M_EPT];
            unsigned short ni[M_EPT];
#pragma unroll
            for (int e = 0; e < M_EPT; e++) {
                const int i = tid + e * NT;
                nv[e] = ~0ull; ni[e] = 0;
                if (i < E) {
                    const u64 a0 = s_h[i];
                    const unsigned short i0 = s_ix[i];
                    const bool has = i + d < E;
                    const u64 a1 = has ? s_h[i + d] : ~0ull;
                    const unsigned short i1 = has ? s_ix[i + d] : i0;
                    const bool right = a1 < a0;  // ties: the left one (leftmost minimum)
                    nv[e] = right ? a1 : a0;
                    ni[e] = right ? i1 : i0;
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < M_EPT; e++) {
                const int i = tid + e * NT;
                if (i < E) { s_h[i] = nv[e]; s_ix[i] = ni[e]; }
            }
            __syncthreads();
        }
    }
    u64 val[MPT], apos[MPT];
    u32 keep = 0;
#pragma unroll
    for (int jj = 0; jj < MPT; jj++) {
        const int i = tid + jj * NT;
        val[jj] = 0; apos[jj] = 0;
        if (!((valid >> jj) & 1u)) continue;
        const u64 g = gpos[jj], last = lasth[jj];
        bool e;
        u64 v, a;
        if (w == 1) {
            e = true; v = last; a = g;
        } else {
            // inner window = s_h[i + 1 .. i + W1]: two table entries of length 2^top that cover it
            const int a0 = i + 1, a1 = i + 1 + W1 - (1 << top);
            const u64 m0 = s_h[a0], m1 = s_h[a1];
            const bool right = m1 < m0;
            const u64 m = right ? m1 : m0;
            const int am = (int)(right ? s_ix[a1] : s_ix[a0]) - a0;
            e = (g == 0) || (prevh[jj] <= m) || (last < m);
            v = last < m ? last : m;
            a = last < m ? g + (u64)w - 1 : g + (u64)am;
        }
        if (e && (p.max_hash == 0 || v <= p.max_hash)) {
            keep |= 1u << jj;
            val[jj] = v; apos[jj] = a;
        }
    }
    u32 before[MPT];
#pragma unroll
    for (int jj = 0; jj < MPT; jj++) {
        const u64 m = __ballot((keep >> jj) & 1u);
        before[jj] = (u32)__popcll(m & ((1ull << lane) - 1));
        if (lane == 0) s_cnt[jj * NWV + wave] = (u32)__popcll(m);
    }
    __syncthreads();
    if (tid < 64) {
        constexpr int NG = MPT * NWV;
        const u32 c = lane < NG ? s_cnt[lane] : 0;
        const u32 incl = wave_incl_scan_u32(c);
        if (lane < NG) s_cnt[lane] = incl - c;
        if (lane == 63) s_cnt[NG] = incl;
    }
    __syncthreads();
    const u32 tile_total = s_cnt[MPT * NWV];
    if (tid < 64) {
        bool timed_out = false;
        if (lane == 0) lb_publish(p.status, tile, (u64)tile_total);
        const u64 base = lb_resolve(p.status, tile, (u64)tile_total, lane, TICKET ? nullptr : &timed_out);
        if (tid == 0) s_misc[1] = base;
        if (timed_out && lane == 0) atomicOr((unsigned long long *)&p.result[1], 2ull);
    }
    __syncthreads();
    const u64 base = s_misc[1];
#pragma unroll
    for (int jj = 0; jj < MPT; jj++)
        if ((keep >> jj) & 1u) {
            const u64 pos = base + s_cnt[jj * NWV + wave] + before[jj];
            if (pos < p.out_cap) {
                p.out[pos] = val[jj];
                if (p.out_pos) p.out_pos[pos] = apos[jj];
            }
        }
    if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
}

int run_minimizer(ukm_ctx *c, const u8 *bases, const u64 *rec_off, u64 n_rec, int k, int w, int circular,
                  u64 max_hash, u64 *out, u64 *out_pos, u64 out_cap, u64 *n_out, u64 total_bases) {
    *n_out = 0;
    if (n_rec == 0 || total_bases == 0) return UKM_OK;
    // every canonical hash, in a workspace buffer (windows <= bases, also when circular)
    u64 *h = nullptr;
    UKM_TRY(ws_alloc_t(c, total_bases, &h));
    u64 n_h = 0;
    const u64 *off = nullptr;
    UKM_TRY(run_windows(c, true, bases, rec_off, n_rec, k, 1, circular, 0, h, total_bases, &n_h, total_bases, &off));
    if (n_h == 0) return UKM_OK;
    const u64 ntiles = (n_h + MT - 1) / MT;
    u64 *ctl = nullptr, *tile_rec = nullptr;
    const size_t nctl = 8 + lb_status_words(ntiles);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
    UKM_TRY(ws_alloc_t(c, ntiles + 3, &tile_rec));
    hipLaunchKernelGGL(tile_first_rec_kernel, dim3((unsigned)((ntiles + 3 + 255) / 256)), dim3(256), 0, c->stream,
                       off, n_rec, n_h, ntiles, (u64)MT, tile_rec);
    MinArgs p;
    memset(&p, 0, sizeof(p));
    p.h = h; p.off = off; p.n_rec = n_rec; p.total = n_h; p.w = w; p.max_hash = max_hash;
    p.out = out; p.out_pos = out_pos; p.out_cap = out_cap;
    p.result = ctl; p.ticket = (u32 *)(ctl + 2); p.status = ctl + 8; p.ntiles = ntiles; p.tile_rec = tile_rec;
    u64 res = 0;
    for (int attempt = c->setop_force_ticket ? 1 : 0; attempt < 2; attempt++) {
        if (attempt == 1) UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
        (void)hipEventRecord(c->ev_k0, c->stream);
        if (attempt == 0) hipLaunchKernelGGL(minimizer_kernel<false>, dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        else hipLaunchKernelGGL(minimizer_kernel<true>, dim3((unsigned)ntiles), dim3(NT), 0, c->stream, p);
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        u64 r2[2];
        UKM_TRY(ukm_read_u64(c, ctl, r2, 2));
        res = r2[0];
        if (!(r2[1] & 2)) break;
        if (attempt == 1) UKM_FAIL(UKM_ERR_HIP, "minimizer kernel: look-back watchdog fired in the ticketed kernel");
        ukm_switch_to_tickets(c, "minimizer kernel");
    }
    *n_out = res;
    if (res > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu values, capacity is %llu", (unsigned long long)res,
                 (unsigned long long)out_cap);
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

// sketches.NewMinimizerSketch(seq, k, w, circular).NextMinimizer() (count.go:316,357) + the Scaled
// filter applied to the emitted minimizers (count.go:373-375)
extern "C" int ukm_minimizer(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec,
                             int k, int w, int circular, uint64_t max_hash, uint64_t *out,
                             uint64_t *out_pos, uint64_t out_cap, uint64_t *n_out) {
    const char *name = "ukm_minimizer";
    if (!ctx || !n_out || (!out && out_cap) || (n_rec && (!rec_off || !bases)))
        UKM_FAIL(UKM_ERR_INVALID, "%s: NULL argument", name);
    if (k < 1 || k > 64) UKM_FAIL(UKM_ERR_K, "%s: k = %d out of range", name, k);
    if (w < 1 || w > MW_MAX) UKM_FAIL(UKM_ERR_INVALID, "%s: w = %d out of range [1, %d]", name, w, MW_MAX);
    *n_out = 0;
    if (n_rec == 0) return UKM_OK;
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *off = nullptr;
        UKM_TRY(ukm_in_t(ctx, rec_off, n_rec + 1, &off));
        u64 total_bases = 0, first = 0;
        if (ukm_is_device_ptr(rec_off)) {
            UKM_TRY(ukm_read_u64(ctx, off + n_rec, &total_bases));
            UKM_TRY(ukm_read_u64(ctx, off, &first));
        } else {
            total_bases = rec_off[n_rec];
            first = rec_off[0];
        }
        if (first != 0) UKM_FAIL(UKM_ERR_INVALID, "%s: rec_off[0] must be 0", name);
        const u8 *b = nullptr;
        u64 *o = nullptr, *op = nullptr;
        UKM_TRY(ukm_in_t(ctx, bases, total_bases, &b));
        UKM_TRY(ukm_out_t(ctx, out, out_cap, &o));
        if (out_pos) UKM_TRY(ukm_out_t(ctx, out_pos, out_cap, &op));
        int r = run_minimizer(ctx, b, off, n_rec, k, w, circular, max_hash, o, op, out_cap, n_out, total_bases);
        ukm_out_resize(ctx, out, (r == UKM_OK ? *n_out : 0) * sizeof(u64));
        if (out_pos) ukm_out_resize(ctx, out_pos, (r == UKM_OK ? *n_out : 0) * sizeof(u64));
        return r;
    }();
    return ukm_finish(&s, rc);
}

// `count` in ONE call: every window (codes, or ntHash with the Scaled filter) -> sort -> the distinct / repeated / singleton
// set, the body of the Run closure count.go:285-436 (iterator, `m[code] = struct{}{}` per k-mer, the `-u` / `-d` marks) and
// its sort count.go:581.  The windows never leave the device: they are produced into the context's workspace, sorted there
// and reduced into `out`; one stream synchronisation and one read-back for the whole call instead of three of each (round-5
// review: the CLI's count_on_device was the caller the fused entry point did not have).
extern "C" int ukm_count(ukm_ctx *ctx, const uint8_t *bases, const uint64_t *rec_off, uint64_t n_rec, int k, int canonical,
                         int circular, int hashed, uint64_t max_hash, int mode, uint64_t *out, uint64_t out_cap, uint64_t *n_out) {
    const char *name = "ukm_count";
    if (!ctx || !n_out || (!out && out_cap) || (n_rec && (!rec_off || !bases))) UKM_FAIL(UKM_ERR_INVALID, "%s: NULL argument", name);
    if (k < 1 || k > (hashed ? 64 : 32)) UKM_FAIL(UKM_ERR_K, "%s: k = %d out of range", name, k);
    if (mode != UKM_UNIQUE && mode != UKM_REPEATED && mode != UKM_SINGLETON)
        UKM_FAIL(UKM_ERR_INVALID, "%s: mode must be UKM_UNIQUE, UKM_REPEATED (-d) or UKM_SINGLETON (-u)", name);
    if (!hashed && max_hash) UKM_FAIL(UKM_ERR_INVALID, "%s: max_hash (--scale) needs hashed = 1", name);
    *n_out = 0;
    if (n_rec == 0) return UKM_OK;
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *off = nullptr;
        UKM_TRY(ukm_in_t(ctx, rec_off, n_rec + 1, &off));
        u64 total_bases = 0, first = 0;
        if (ukm_is_device_ptr(rec_off)) {
            UKM_TRY(ukm_read_u64(ctx, off + n_rec, &total_bases));
            UKM_TRY(ukm_read_u64(ctx, off, &first));
        } else {
            total_bases = rec_off[n_rec];
            first = rec_off[0];
        }
        if (first != 0) UKM_FAIL(UKM_ERR_INVALID, "%s: rec_off[0] must be 0", name);
        const u8 *b = nullptr;
        u64 *o = nullptr;
        UKM_TRY(ukm_in_t(ctx, bases, total_bases, &b));
        UKM_TRY(ukm_out_t(ctx, out, out_cap, &o));
        // windows <= bases (circular records: one per base); with a Scaled filter a small share of them
        u64 wcap = total_bases + 1;
        if (hashed && max_hash && max_hash != ~0ull) {
            const double keep = 2.0 * ((double)max_hash / 18446744073709551615.0);  // canonical = the smaller of two hashes
            wcap = std::min<u64>(wcap, (u64)((double)total_bases * std::min(1.0, 1.5 * keep)) + (1u << 20));
        }
        u64 *w = nullptr;
        UKM_TRY(ws_alloc_t(ctx, (size_t)wcap, &w));
        int bits = hashed ? 64 : 2 * k;
        if (hashed && max_hash && max_hash != ~0ull) bits = 64 - __builtin_clzll(max_hash);
        // the sort's first histogram is counted by the kernel that writes the windows (the strip kernel, when it runs)
        FusedHist fh = {nullptr, bits, -1};
        UKM_TRY(ws_alloc_t(ctx, 256, &fh.hist));
        UKM_HIP(hipMemsetAsync(fh.hist, 0, 256 * sizeof(u64), ctx->stream));
        u64 nw = 0;
        UKM_TRY(run_windows(ctx, hashed != 0, b, off, n_rec, k, canonical, circular, max_hash, w, wcap, &nw, total_bases, nullptr, &fh));
        if (nw == 0) {
            ukm_out_resize(ctx, out, 0);
            return UKM_OK;
        }
        UKM_TRY(ukm_dev_sort_hist(ctx, w, nullptr, nw, bits, fh.shift >= 0 ? fh.hist : nullptr, fh.shift));
        int r = ukm_dev_unique(ctx, w, nullptr, nw, mode, o, nullptr, out_cap, n_out);
        ukm_out_resize(ctx, out, (r == UKM_OK ? *n_out : 0) * sizeof(u64));
        return r;
    }();
    return ukm_finish(&s, rc);
}

// count.go:98  maxHash := uint64(float64(^uint64(0)) / float64(scale))
extern "C" uint64_t ukm_max_hash(uint64_t scale) {
    if (scale <= 1) return ~0ull;
    double d = 18446744073709551615.0 / (double)scale;  // float64(^uint64(0)) rounds to 2^64
    return (uint64_t)d;
}
