// ukm_sort.hip — LSD radix sort of uint64 codes (optionally carrying uint32 taxids), the
// MI355X replacement for sortutil.Uint64s / sorts.Quicksort(CodeTaxidSlice)
// (twotwotwo/sorts; call sites count.go:581, union.go:274,295, sort.go:268,331,457,463 ...).
//
// Design (HBM-bound, integer):
//   * one histogram kernel reads the keys once and builds the 256-bin digit histograms of ALL
//     passes (LDS atomics, with a wave-uniform fast path so sorted/skewed input does not
//     serialise on one bin); the host turns them into per-pass digit bases and drops passes
//     whose digit is constant (k=31 -> 62 significant bits; top bits of k=21 codes are zero);
//   * per executed pass ONE "onesweep" kernel: each 256-thread workgroup takes a ticketed tile
//     of 7168 keys (512 threads x 14; 11264 = 1024 x 11 with taxids), ranks them stably with wave64 ballot match-any (8 ballots per key) into
//     per-wave LDS digit counters, resolves the tile's global digit offsets by a per-digit
//     decoupled look-back over the previous tiles' counts (thread d owns digit d), reorders
//     the tile through LDS so that global stores are contiguous per digit, and scatters.
//   Algorithmic bytes: 8n (histogram) + P * (8n read + 8n write) [+ P * 8n for taxids].
#include <algorithm>
#include <utility>
#include <vector>
#include <cmath>

#include "ukm_device.h"

namespace {

// onesweep tile shapes (threads x keys per thread), measured on MI355X at 1e8 keys (profiles/r01_notes.md, r02_notes.md):
// keys only 1024x14 with the fused next-digit histogram (3.67 ms; round 1 without it: 512x14 4.15, 512x12 4.3,
// 512x15 5.3, 256x24 4.6, 768x10 4.9, 1024x7 5.1), key + taxid 1024x11 (5.1 ms; 768x14 5.1, 512x20 5.4, 512x12 6.4)
#ifndef SORT_NT_KEYS
#define SORT_NT_KEYS 1024
#endif
#ifndef SORT_VT_KEYS
#define SORT_VT_KEYS 14
#endif
#ifndef SORT_NT_PAIRS
#define SORT_NT_PAIRS 1024
#endif
#ifndef SORT_VT_PAIRS
#define SORT_VT_PAIRS 11
#endif
constexpr int NT = 256;  // histogram kernel
#ifndef SORT_RADIX_BITS
#define SORT_RADIX_BITS 8
#endif
constexpr int RB = SORT_RADIX_BITS;  // digit width: 8 -> 8 passes for 62..64 bits; 9 -> 7 passes for k=31 codes (62 bits), 5 for k=21
constexpr int RADIX = 1 << RB;
constexpr u32 DMASK = RADIX - 1;
constexpr int MAX_PASSES = 8;

// ---- histogram of all digits -------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) U2a8 {  // 16 bytes at 8-byte alignment
    u64 x, y;
};
__global__ __launch_bounds__(NT) void radix_hist_kernel(const u64 *k, u64 n, int passes, u64 *ghist, u64 *or_all, int shift0 = 0) {
    __shared__ u32 s_h[MAX_PASSES * RADIX];
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < MAX_PASSES * RADIX; i += NT) s_h[i] = 0;
    __syncthreads();
    constexpr int U = 4;  // independent 16-byte loads in flight per thread (one 8-byte load per iteration was latency
                          // bound: 1.6 TB/s; four: 3.6 TB/s; four pairs: 4.3 TB/s.  Four copies of every bin, chosen by the lane, were slower: 0.206 against 0.185 ms)
    constexpr u64 STEP = (u64)NT * U * 2;
    const u64 per_block = ((n + gridDim.x - 1) / gridDim.x + STEP - 1) / STEP * STEP;
    const u64 beg = (u64)blockIdx.x * per_block;
    const u64 end = (beg + per_block < n) ? beg + per_block : n;
    u64 orv = 0;  // OR of every key this thread sees: tells the host which high digits are all zero
    for (u64 i0 = beg; i0 < end; i0 += STEP) {
        u64 key[2 * U];
        bool valid[2 * U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + ((u64)u * NT + tid) * 2;
            valid[2 * u] = i < end;
            valid[2 * u + 1] = i + 1 < end;
            if (i + 1 < end) {
                const U2a8 q = *reinterpret_cast<const U2a8 *>(k + i);  // (the caller's array may start on an odd 8-byte slot)
                key[2 * u] = q.x;
                key[2 * u + 1] = q.y;
            } else {
                key[2 * u] = k[valid[2 * u] ? i : end - 1];
                key[2 * u + 1] = key[2 * u];
            }
        }
#pragma unroll
        for (int u = 0; u < 2 * U; u++) {
            const u64 vm = __ballot(valid[u]);
            if (valid[u]) orv |= key[u];
            for (int p = 0; p < passes; p++) {
                const u32 d = (u32)(key[u] >> (shift0 + RB * p)) & DMASK;
                const u32 d0 = __builtin_amdgcn_readfirstlane(d);
                if (vm == ~0ull && __all(d == d0)) {
                    if (lane_id() == 0) atomicAdd(&s_h[p * RADIX + d0], 64u);
                } else if (valid[u]) {
                    atomicAdd(&s_h[p * RADIX + d], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < passes * RADIX; i += NT) {
        u32 v = s_h[i];
        if (v) atomicAdd((unsigned long long *)&ghist[i], (unsigned long long)v);
    }
    if (or_all) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) orv |= __shfl_xor(orv, d, 64);
        if (lane_id() == 0 && orv) atomicOr((unsigned long long *)or_all, (unsigned long long)orv);
    }
}

// ---- onesweep pass -------------------------------------------------------------------------------
template <typename SW> struct SWTraits;
template <> struct SWTraits<u32> {
    static constexpr u32 AGG = 1u << 30, INCL = 2u << 30, VAL = (1u << 30) - 1;
    static constexpr int SHIFT = 30;
};
template <> struct SWTraits<u64> {
    static constexpr u64 AGG = 1ull << 62, INCL = 2ull << 62, VAL = (1ull << 62) - 1;
    static constexpr int SHIFT = 62;
};

template <typename SW>
struct PassArgs {
    const u64 *kin;
    u64 *kout;
    const u32 *vin;
    u32 *vout;
    u64 n;
    int shift;
    SW *status;  // [ntiles][256]
    u32 *ticket;
    u32 *flags;        // bit0: look-back watchdog fired
    const u64 *gbase;  // [256] exclusive digit bases of this pass
    u64 ntiles;
    u64 *next_hist;    // fused mode: [256] digit counts of the NEXT pass, accumulated while this pass scatters (or nullptr)
    int next_shift;
};

// exclusive digit bases of one pass from its histogram (fused mode: runs on the stream between two passes, so the
// host never sees the histograms)
__global__ void radix_bases_kernel(const u64 *hist, u64 *gbase) {
    __shared__ u64 s[RADIX];
    const int d = (int)threadIdx.x;
    s[d] = hist[d];
    __syncthreads();
    if (d == 0) {
        u64 sum = 0;
        for (int i = 0; i < RADIX; i++) {
            const u64 v = s[i];
            s[i] = sum;
            sum += v;
        }
    }
    __syncthreads();
    gbase[d] = s[d];
}

// "match any" on an 8-bit digit: on return (phi:plo) is the 64-bit mask of the lanes whose digit
// equals this lane's.  Per bit: sign-extended bit x (0 / -1), ballot m = (x != 0), peers &= ~(m ^ x)
// with gfx950's three-input v_bitop3_b32 (truth table 0x90 = a & ~(b ^ c)) on each 32-bit half:
// 4 VALU instructions per bit.  Hand-scheduled because a VALU read of an SGPR needs 2 wait states
// after the VALU write: three ballot registers rotate so that every v_cmp is at least 2
// instructions ahead of its first reader (the compiler's version spent s_nops or extra shifts
// here and the ranking made the kernel VALU bound: 85 -> 36 instructions per key).
__device__ __forceinline__ void match_any8(u32 d, u32 &plo, u32 &phi) {
    u32 x, y, z;
    plo = ~0u;
    phi = ~0u;
    asm("v_bfe_i32 %2, %5, 0, 1\n\t"
        "v_bfe_i32 %3, %5, 1, 1\n\t"
        "v_bfe_i32 %4, %5, 2, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_cmp_ne_u32_e64 s[80:81], 0, %3\n\t"
        "v_cmp_ne_u32_e64 s[82:83], 0, %4\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bfe_i32 %2, %5, 3, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_bitop3_b32 %0, %0, s80, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s81, %3 bitop3:0x90\n\t"
        "v_bfe_i32 %3, %5, 4, 1\n\t"
        "v_cmp_ne_u32_e64 s[80:81], 0, %3\n\t"
        "v_bitop3_b32 %0, %0, s82, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s83, %4 bitop3:0x90\n\t"
        "v_bfe_i32 %4, %5, 5, 1\n\t"
        "v_cmp_ne_u32_e64 s[82:83], 0, %4\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bfe_i32 %2, %5, 6, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_bitop3_b32 %0, %0, s80, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s81, %3 bitop3:0x90\n\t"
        "v_bfe_i32 %3, %5, 7, 1\n\t"
        "v_cmp_ne_u32_e64 s[80:81], 0, %3\n\t"
        "v_bitop3_b32 %0, %0, s82, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s83, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s80, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s81, %3 bitop3:0x90"
        : "+v"(plo), "+v"(phi), "=&v"(x), "=&v"(y), "=&v"(z)
        : "v"(d)
        : "vcc", "s80", "s81", "s82", "s83");  // ordinary SGPRs: the top ones (s100, s101 ...) are reserved by the compiler
}

// 9-bit digits: the eight low bits through match_any8, the ninth with one more bfe / cmp / 2 x bitop3 step
__device__ __forceinline__ void match_any_digit(u32 d, u32 &plo, u32 &phi) {
    match_any8(d, plo, phi);
    if (RB == 9) {
        int sb = __builtin_amdgcn_sbfe((int)d, 8u, 1u);
        const u64 m = __ballot(sb != 0);
        plo = __builtin_amdgcn_bitop3_b32(plo, (u32)m, (u32)sb, 0x90);
        phi = __builtin_amdgcn_bitop3_b32(phi, (u32)(m >> 32), (u32)sb, 0x90);
    }
}

#ifndef SORT_LB_W
#define SORT_LB_W 4
#endif
#ifndef SORT_TICKET
#define SORT_TICKET 1
#endif
// TICKET = false: tile id = blockIdx.x (see ukm_setops.hip for the liveness argument and the
// watchdog); TICKET = true: ids from an atomic counter, dispatch-order independent.
#ifdef SORT_WAVES_PER_EU
#define SORT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(SORT_WAVES_PER_EU, SORT_WAVES_PER_EU)))
#else
#define SORT_WAVES_ATTR
#endif
template <typename SW, bool PAIRS, bool TICKET, int NT_, int VT_>
__global__ __launch_bounds__(NT_) SORT_WAVES_ATTR void onesweep_kernel(PassArgs<SW> p) {
    constexpr int NT = NT_, NW = NT_ / 64, VT = VT_, TILE = NT_ * VT_;  // NT >= RADIX: thread d < 256 owns digit d
    static_assert(NT_ >= RADIX && NT_ % 64 == 0, "workgroup must cover all digits");
    static_assert(64 * VT_ < 65536, "per-wave digit counts are 16-bit");
    using T = SWTraits<SW>;
    __shared__ u64 s_keys[TILE];
    __shared__ u32 s_vals[PAIRS ? TILE : 1];
    __shared__ unsigned short s_whist[NW][RADIX];  // per-wave digit counts (<= 64 * VT)
    __shared__ u32 s_dexcl[RADIX];
    __shared__ u64 s_gbase[RADIX];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_tile;
    __shared__ u32 s_nh[RADIX];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (TICKET && tid == 0) s_tile = atomicAdd(p.ticket, 1u);
    if (tid < RADIX) s_nh[tid] = 0;
    for (int i = tid; i < NW * RADIX / 2; i += NT) reinterpret_cast<u32 *>(&s_whist[0][0])[i] = 0;
    __syncthreads();
    const u64 tile = TICKET ? (u64)s_tile : (u64)blockIdx.x;
    const u64 tbase = tile * (u64)TILE;
    const u32 valid_count = (u32)((p.n - tbase < (u64)TILE) ? (p.n - tbase) : (u64)TILE);

    // wave w owns keys [w*64*VT, (w+1)*64*VT) of the tile, striped across its lanes
    u64 key[VT];
    u32 val[VT];
    u32 rank[VT];
    const u32 wbase_idx = (u32)wave * 64 * VT + (u32)lane;
#pragma unroll
    for (int j = 0; j < VT; j++) {
        const u32 li = wbase_idx + j * 64;
        const bool v = li < valid_count;
        key[j] = v ? p.kin[tbase + li] : ~0ull;
        if (PAIRS) val[j] = v ? p.vin[tbase + li] : 0;
    }
    // Stable ranking inside the wave by "match any": after 8 ballots `peers` holds the lanes whose
    // digit equals this lane's.  Written on 32-bit halves with a sign-extended bit (0 / -1) and gfx950's
    // three-input v_bitop3_b32 (peers & ~(ballot ^ bit) in ONE instruction), so that a digit bit costs
    // v_bfe_i32 + v_cmp + 2 x v_bitop3: the 64-bit `bit ? m : ~m` form compiled to 9 VALU
    // instructions per bit and made the kernel VALU bound.
    const u32 lt_lo = lane < 32 ? ((1u << lane) - 1u) : ~0u;
    const u32 lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
#pragma unroll
    for (int j = 0; j < VT; j++) {
        const u32 li = wbase_idx + j * 64;
        // padding items take the highest digit; they sit at the end of the tile order, so they rank
        // after every real key of that digit and are dropped at write-out
        const u32 d = (li < valid_count) ? ((u32)(key[j] >> p.shift) & DMASK) : DMASK;
        u32 plo, phi;
        match_any_digit(d, plo, phi);
        const u32 pre = s_whist[wave][d];
        const u32 r = (u32)__popc(plo & lt_lo) + (u32)__popc(phi & lt_hi);
        const u32 tot = (u32)__popc(plo) + (u32)__popc(phi);
        rank[j] = pre + r;
        // fused mode: this key's digit of the NEXT pass is counted here, so that no separate pass over the keys is
        // needed for it (the pre-pass builds the first histogram only)
        if (p.next_hist && li < valid_count) atomicAdd(&s_nh[(u32)(key[j] >> p.next_shift) & DMASK], 1u);
        // every peer stores the same new count (same address, same value): no branch, so the 16 keys'
        // ranking stays one basic block that the scheduler can interleave
        s_whist[wave][d] = (unsigned short)(pre + tot);
    }
    __syncthreads();

    // thread d: exclusive scan over the waves, tile count of digit d
    const int d = tid & (RADIX - 1);
    const bool owner = tid < RADIX;  // workgroup-size independent: the first 256 threads own the digits
    u32 cnt = 0;
    if (owner) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
            u32 c = s_whist[w][d];
            s_whist[w][d] = (unsigned short)cnt;
            cnt += c;
        }
    }
    u32 real_cnt = cnt;
    if (d == (int)DMASK) real_cnt -= (u32)TILE - valid_count;
    // publish the tile's count of digit d, then look back
    SW *st = p.status + tile * RADIX + d;
    if (owner) {
        if (tile == 0) __hip_atomic_store(st, (SW)(T::INCL | (SW)real_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(st, (SW)(T::AGG | (SW)real_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    u32 tile_total;
    const u32 dex = block_excl_scan_u32<NT>(cnt, s_scan, &tile_total);  // contains barriers
    if (owner) s_dexcl[d] = dex;
    __syncthreads();

    // local reorder: tile becomes digit-sorted (stable) in LDS
#pragma unroll
    for (int j = 0; j < VT; j++) {
        const u32 li = wbase_idx + j * 64;
        const u32 dd = (li < valid_count) ? ((u32)(key[j] >> p.shift) & DMASK) : DMASK;
        const u32 pos = s_dexcl[dd] + s_whist[wave][dd] + rank[j];
        s_keys[pos] = key[j];
        if (PAIRS) s_vals[pos] = val[j];
    }

    // per-digit decoupled look-back: thread d walks back over the tiles' counts of digit d,
    // SORT_LB_W tiles per hop (independent loads in flight together; one hop is a ~1.5 us
    // device-scope round trip, so a one-tile-per-hop walk was latency bound)
    u64 excl = 0;
    bool timed_out = false;
#ifdef SORT_ABL_NOLB  // experiment only: plausible but wrong offsets, no look-back
    excl = tile * (u64)real_cnt;
    if (p.gbase[d] + excl + real_cnt > p.n) excl = 0;
    if (false) {
#else
    if (tile > 0 && owner) {
#endif
        long long t = (long long)tile - 1;
        u32 spins = 0;
        bool done = false;
        while (!done) {
            SW w[SORT_LB_W];
#pragma unroll
            for (int q = 0; q < SORT_LB_W; q++) {
                const long long tq = t - q;
                w[q] = (tq >= 0) ? __hip_atomic_load(p.status + (u64)tq * RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : (SW)T::INCL;
            }
            int used = 0;
#pragma unroll
            for (int q = 0; q < SORT_LB_W; q++) {
                if (!done && used == q) {
                    const u32 state = (u32)(w[q] >> T::SHIFT);
                    if (state != 0) {
                        excl += (u64)(w[q] & T::VAL);
                        used = q + 1;
                        if (state == 2) done = true;
                    }
                }
            }
            t -= used;
            if (!done && used < SORT_LB_W) {  // hit an unpublished tile: back off, then re-read from it
                // (ticketed tile ids: every predecessor is running, the wait always ends — no watchdog, the host
                //  does not look at the flag in that mode)
                if (!TICKET && ++spins > LB_SPIN_LIMIT) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __hip_atomic_store(st, (SW)(T::INCL | (SW)(excl + real_cnt)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (timed_out) atomicOr(p.flags, 1u);
    if (owner) s_gbase[d] = p.gbase[d] + excl - (u64)dex;
    if (p.next_hist && owner) {
        const u32 c = s_nh[d];  // (complete: at least one barrier lies between the counting and here)
        if (c) atomicAdd((unsigned long long *)&p.next_hist[d], (unsigned long long)c);
    }
    __syncthreads();

    for (u32 i = (u32)tid; i < valid_count; i += NT) {
        const u64 kk = s_keys[i];
        const u32 dd = (u32)(kk >> p.shift) & DMASK;
        const u64 pos = s_gbase[dd] + i;
        p.kout[pos] = kk;
        if (PAIRS) p.vout[pos] = s_vals[i];
    }
}

template <typename SW, bool PAIRS, int NT_, int VT_>
int run_passes(ukm_ctx *c, u64 *keys, u32 *vals, u64 *tk, u32 *tv, u64 n, int npass,
               const int *shifts, const u64 *gbase_dev, bool *result_in_tmp, u64 *fused_hist = nullptr) {
    constexpr int TILE = NT_ * VT_;
    const u64 ntiles = (n + TILE - 1) / TILE;
    SW *status = nullptr;
    u64 *ctl = nullptr;  // [0] ticket, [1] flags
    UKM_TRY(ws_alloc_t(c, ntiles * RADIX, &status));
    UKM_TRY(ws_alloc_t(c, 2, &ctl));
    u64 *src_k = keys, *dst_k = tk;
    u32 *src_v = vals, *dst_v = tv;
    for (int i = 0; i < npass; i++) {
        // Pass i.  The blockIdx-ordered variant is checked per pass (its source buffer is still
        // intact if the watchdog fired, so only that pass is repeated with tickets).
        for (int attempt = (c->setop_force_ticket || SORT_TICKET) ? 1 : 0; attempt < 2; attempt++) {
            const bool ticket = attempt == 1;
            UKM_HIP(hipMemsetAsync(status, 0, ntiles * RADIX * sizeof(SW), c->stream));
            UKM_HIP(hipMemsetAsync(ctl, 0, 2 * sizeof(u64), c->stream));
            PassArgs<SW> p;
            p.kin = src_k; p.kout = dst_k; p.vin = src_v; p.vout = dst_v;
            p.n = n; p.shift = shifts[i];
            p.status = status; p.ticket = (u32 *)ctl; p.flags = (u32 *)(ctl + 1);
            p.gbase = gbase_dev + (size_t)i * RADIX;
            p.ntiles = ntiles;
            p.next_hist = nullptr;
            p.next_shift = 0;
            if (fused_hist) {
                // bases of this pass from its histogram (built by the pre-pass for i = 0, by pass i - 1 otherwise)
                if (attempt == ((c->setop_force_ticket || SORT_TICKET) ? 1 : 0))
                    hipLaunchKernelGGL(radix_bases_kernel, dim3(1), dim3(RADIX), 0, c->stream, fused_hist + (size_t)i * RADIX,
                                       const_cast<u64 *>(p.gbase));
                if (i + 1 < npass) {
                    p.next_hist = fused_hist + (size_t)(i + 1) * RADIX;
                    p.next_shift = shifts[i + 1];
                }
            }
            const dim3 grid((unsigned)ntiles), block(NT_);
            if (ticket) hipLaunchKernelGGL((onesweep_kernel<SW, PAIRS, true, NT_, VT_>), grid, block, 0, c->stream, p);
            else hipLaunchKernelGGL((onesweep_kernel<SW, PAIRS, false, NT_, VT_>), grid, block, 0, c->stream, p);
            UKM_HIP(hipGetLastError());
            if (ticket) break;  // cannot stall
            u64 fl = 0;
            UKM_TRY(ukm_read_u64(c, ctl + 1, &fl));
            if (!(fl & 1)) break;
            ukm_switch_to_tickets(c, "radix sort pass");  // this device does not dispatch in order
            if (fused_hist && i + 1 < npass)  // the repeated pass counts the next digit again
                UKM_HIP(hipMemsetAsync(fused_hist + (size_t)(i + 1) * RADIX, 0, RADIX * sizeof(u64), c->stream));
        }
        std::swap(src_k, dst_k);
        std::swap(src_v, dst_v);
    }
    *result_in_tmp = (src_k != keys);
    return UKM_OK;
}

// ---- two passes over the TOP 16 bits, then every bucket sorted inside LDS -------------------------------------------
// A scatter pass costs ~0.43 ms per 1e8 keys whatever it moves (latency bound, above), and a 62-bit key takes eight of
// them.  Sorted by its top 16 bits alone (two passes: LSD over the two highest digits) the array falls into 65,536
// buckets of n / 65,536 keys on evenly spread keys (k-mer codes, hashes): 1.5 k keys = 12 KB at n = 1e8, which one
// 256-thread workgroup sorts by the remaining low bits entirely in LDS (the same ballot ranking as the global pass,
// ping-pong between two LDS buffers) and writes back in place.  Keys cross HBM three times instead of eight.
// Twelve instantiations (256 ... 4096 keys) serve the buckets by size class; buckets beyond 4096 keys are sorted by the
// general route afterwards; too many of those, narrow keys, or n beyond ~1.3e8 and the whole call takes the general
// route (the two passes only permuted the keys).
constexpr int LS_NT = 256, LS_NW = LS_NT / 64;  // keys per thread 2 / 4 / 8 / 16: buckets of up to 512 ... 4096 keys
constexpr int LS_TOP_MIN = 12, LS_TOP_MAX = 22;  // bucket = the top `topb` bits: 2^topb buckets of ~1400 keys
constexpr int LS_MAX_BIG = 8192;  // buckets beyond 4096 keys, gathered side by side and sorted together by the general route
constexpr int LS_NCLASS = 12;
constexpr int LS_CLASS_KPT[LS_NCLASS] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};  // keys per thread of the size classes: 256 ... 4096 keys

struct LocalSortArgs {
    const u64 *src;    // sorted by its top bits (the caller's array after two top passes, the scratch copy after three)
    const u32 *vsrc;   // taxids riding along (PAIRS) or nullptr
    u64 *keys;         // the caller's array: every bucket is written to its own segment
    u32 *vals;
    u64 n;
    const u64 *start;  // [2^topb + 1]
    int low_bits;      // bits below the bucket bits
    const u32 *ids;    // the buckets of this launch's size class
    int counting;      // 1: one counting step + a walk inside the bins (round 5); 0: digit passes only
    u64 *stat;         // [0] += 1 per bucket that fell back from the counting step
};

// start[b] = first index whose bucket value (key >> low_bits) is >= b (b = 0 .. 2^topb)
__global__ void ls_bounds_kernel(const u64 *keys, u64 n, int low_bits, int topb, u64 *start) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > (1u << topb)) return;
    u64 lo = 0, hi = n;
    if (b == (1u << topb)) {
        lo = n;
    } else {
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if ((keys[mid] >> low_bits) < (u64)b) lo = mid + 1; else hi = mid;
        }
    }
    start[b] = lo;
}

// A sample of 2^20 keys counted by bucket, then the buckets whose share of the sample says "beyond 4096 keys":
// run before the two passes on a context that has already met crowded keys (a wasted attempt costs two passes)
__global__ void ls_sample_kernel(const u64 *keys, u64 n, int low_bits, int topb, u32 *cnt, u32 nsamp) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamp) return;
    const u64 pos = (u64)(((unsigned __int128)i * n) / nsamp);
    const u64 bb = keys[pos] >> low_bits;
    const u32 b = bb < (1u << topb) ? (u32)bb : (1u << topb) - 1;
    // one atomic per distinct bucket of the wave (crowded keys: 2^20 atomics on a few dozen addresses took ~1 ms)
    bool todo = true;
    while (todo) {
        const u32 b0 = (u32)__builtin_amdgcn_readfirstlane((int)b);
        const u64 same = __ballot(b == b0);
        if (b == b0) {
            if ((int)lane_id() == __ffsll((long long)same) - 1) atomicAdd(&cnt[b0], (u32)__popcll(same));
            todo = false;
        }
    }
}
__global__ void ls_sample_count_kernel(const u32 *cnt, u32 thr, u64 *out) {  // out[0] heavy buckets, out[1] samples in them
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 cb = cnt[b];
    const u64 m = __ballot(cb > thr);
    if (m == 0ull) return;
    u64 in_heavy = cb > thr ? cb : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) in_heavy += __shfl_xor(in_heavy, d, 64);
    if (lane_id() == 0) {
        atomicAdd((unsigned long long *)&out[0], (unsigned long long)__popcll(m));
        atomicAdd((unsigned long long *)&out[1], (unsigned long long)in_heavy);
    }
}

// size class of every bucket: cls[k] = number of buckets of class k (k = LS_NCLASS: beyond every class, their ids in
// cls[LS_NCLASS + 1 ...]); ids[k][...] = the buckets of class k
__global__ void ls_classify_kernel(const u64 *start, u64 *cls, u32 *ids, int topb, u32 *big_ids, u64 *big_size, u64 *stat) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;  // (the grid covers the buckets exactly)
    if (b == 0) cls[LS_NCLASS + 2] = (u64)atomicExch((unsigned long long *)stat, 0ull);  // the previous call's fall-backs
    const u64 m = start[b + 1] - start[b];
    int k = -1;
    if (m > 0) {  // (a bucket of one key still has to reach the caller's array when the sorted-by-top-bits copy is the scratch)
        k = 0;
        while (k < LS_NCLASS && m > (u64)LS_NT * LS_CLASS_KPT[k]) k++;
    }
    // one atomic per wave and class (65,536 atomics on two or three addresses took 0.55 ms)
    const u64 lt = (1ull << lane_id()) - 1ull;
    for (int q = 0; q <= LS_NCLASS; q++) {
        const u64 mask = __ballot(k == q);
        if (mask == 0ull) continue;
        const int lead = __ffsll((long long)mask) - 1;
        u64 at = 0;
        if (lane_id() == lead) at = atomicAdd((unsigned long long *)&cls[q], (unsigned long long)__popcll(mask));
        at = __shfl(at, lead, 64) + (u64)__popcll(mask & lt);
        if (k == q) {
            if (q < LS_NCLASS) ids[(size_t)q << topb | at] = b;
            else if (at < (u64)LS_MAX_BIG) big_ids[at] = b;
        }
    }
    // keys in oversized buckets (cls[LS_NCLASS + 1]): one atomic per wave; big_size[b] = the bucket's size if it is one of
    // them, else 0: its exclusive scan is where every oversized bucket lies in the gathered array (bucket order)
    u64 big = k == LS_NCLASS ? m : 0;
    big_size[b] = big;
    if (__ballot(big != 0) != 0ull) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) big += __shfl_xor(big, d, 64);
        if (lane_id() == 0) atomicAdd((unsigned long long *)&cls[LS_NCLASS + 1], (unsigned long long)big);
    }
}

// the oversized buckets <-> one array in which they lie side by side in bucket order (BACK: the sorted array to the
// caller's).  blockIdx.x = entry of the list, blockIdx.y = one of gridDim.y parts of the bucket (a single code with
// millions of copies -- poly-A -- is one bucket)
template <bool BACK>
__global__ void ls_big_copy_kernel(const u32 *big_ids, const u64 *start, const u64 *big_off, const u64 *src, const u32 *vsrc,
                                   u64 *dst, u32 *vdst) {
    const u32 b = big_ids[blockIdx.x];
    const u64 s0 = start[b], m = start[b + 1] - s0, o = big_off[b];
    const u64 lo = m * blockIdx.y / gridDim.y, hi = m * (blockIdx.y + 1) / gridDim.y;
    for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        if (BACK) {
            dst[s0 + i] = src[o + i];
            if (vsrc) vdst[s0 + i] = vsrc[o + i];
        } else {
            dst[o + i] = src[s0 + i];
            if (vsrc) vdst[o + i] = vsrc[s0 + i];
        }
    }
}

// Round 5: ONE counting step instead of the six digit passes.  The bucket's keys are spread over NBIN >= capacity bins
// by the bits right below the bucket bits (an LDS atomic per key: its arrival number inside the bin), the bin sizes are
// scanned, every key is put into its bin's range, and its place INSIDE the bin is the number of the bin's keys that
// precede it -- a walk over 1.4 keys on average for evenly spread low bits (k-mer codes, hashes), instead of five more
// passes of 8 ballots, two counter updates, four barriers and an LDS round trip each.  Equal keys are ordered by their
// place in the input (pairs: the sort stays stable; plain keys: any order).  A bucket with a bin of more than LS_BIN_MAX
// keys (a repeated k-mer, keys that differ in their lowest bits only) takes the digit passes as before.
constexpr u32 LS_BIN_MAX = 64;
#ifndef LS_WALK_MAX
#define LS_WALK_MAX 4u  /* keys read by the walks, per key of the bucket, beyond which the bucket takes the digit passes (1.75 on evenly spread keys) */
#endif
template <int CAP> struct LsBins {
    static constexpr int NBIN = CAP <= 256 ? 256 : CAP <= 512 ? 512 : CAP <= 1024 ? 1024 : CAP <= 2048 ? 2048 : 4096;
    static constexpr int HB = CAP <= 256 ? 8 : CAP <= 512 ? 9 : CAP <= 1024 ? 10 : CAP <= 2048 ? 11 : 12;
};

template <int LS_KPT, bool PAIRS = false>
__global__ __launch_bounds__(LS_NT) void ls_sort_kernel(LocalSortArgs a) {
    constexpr int LS_CAP = LS_NT * LS_KPT;
    constexpr int NBIN = LsBins<LS_CAP>::NBIN, HB = LsBins<LS_CAP>::HB, BPT = NBIN / LS_NT;
    static_assert((NBIN + 1) * 4 <= LS_CAP * 8, "the bin counters live in the second key buffer");
    static_assert(LS_CAP <= 65536, "16-bit input positions");
    __shared__ __attribute__((aligned(16))) u64 s_buf[2][LS_CAP];
    __shared__ u32 s_vbuf[PAIRS ? 2 : 1][PAIRS ? LS_CAP : 1];
    __shared__ unsigned short s_wh[LS_NW][RADIX];
    __shared__ u32 s_dex[RADIX];
    __shared__ u32 s_scan[LS_NW + 1];
    static_assert(LS_NT == RADIX, "thread d owns digit d");
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const u32 b = a.ids[blockIdx.x];
    const u64 beg = a.start[b], end = a.start[b + 1];
    const u64 m64 = end - beg;
    const u32 m = (u32)m64;
    u64 key[LS_KPT];
    u32 val[PAIRS ? LS_KPT : 1];
    const u32 wbase = (u32)wave * 64 * LS_KPT + (u32)lane;
#pragma unroll
    for (int j = 0; j < LS_KPT; j++) {
        const u32 i = wbase + j * 64;
        key[j] = i < m ? a.src[beg + i] : ~0ull;  // padding: the highest digit in every pass, last in tile order
        if (PAIRS) val[j] = i < m ? a.vsrc[beg + i] : 0u;
    }
    if (a.counting && a.low_bits >= HB) {
        u32 *s_cnt = reinterpret_cast<u32 *>(&s_buf[1][0]);                         // [NBIN + 1]
        unsigned short *s_idx = reinterpret_cast<unsigned short *>(&s_vbuf[PAIRS ? 1 : 0][0]);  // (pairs) input position of the key at a place
        u64 *grp = &s_buf[0][0];
        const int bshift = a.low_bits - HB;
        for (int i = tid; i < NBIN + 1; i += LS_NT) s_cnt[i] = 0;
        __syncthreads();
        u32 pos[LS_KPT];
#pragma unroll
        for (int j = 0; j < LS_KPT; j++) {
            const u32 bin = (u32)(key[j] >> bshift) & (u32)(NBIN - 1);
            pos[j] = 0;
            if (wbase + j * 64 < m) pos[j] = atomicAdd(&s_cnt[bin], 1u);
        }
        __syncthreads();
        u32 c[BPT], sum = 0, mx = 0, sq = 0;
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            c[q] = s_cnt[tid * BPT + q];
            sum += c[q];
            mx = c[q] > mx ? c[q] : mx;
            sq += c[q] * c[q];  // the walks below read (bin size)^2 keys per bin
        }
        u32 total = 0;
        u32 run = block_excl_scan_u32<LS_NT>(sum, s_scan, &total);
        const u32 wsq = wave_incl_scan_u32(sq);
        if (lane == 63) s_scan[wave] = wsq;
        int heavy = __syncthreads_or(mx > LS_BIN_MAX);
        u32 walk = 0;
#pragma unroll
        for (int w = 0; w < LS_NW; w++) walk += s_scan[w];
        heavy |= walk > LS_WALK_MAX * m + 512u;  // (workgroup-uniform) e.g. every key four times: the digit passes are cheaper
        if (heavy && tid == 0) atomicAdd((unsigned long long *)a.stat, 1ull);
        if (!heavy) {
#pragma unroll
            for (int q = 0; q < BPT; q++) {
                s_cnt[tid * BPT + q] = run;
                run += c[q];
            }
            if (tid == LS_NT - 1) s_cnt[NBIN] = run;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LS_KPT; j++) {
                const u32 i = wbase + j * 64;
                if (i < m) {
                    const u32 bin = (u32)(key[j] >> bshift) & (u32)(NBIN - 1);
                    pos[j] += s_cnt[bin];
                    grp[pos[j]] = key[j];
                    if (PAIRS) s_idx[pos[j]] = (unsigned short)i;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LS_KPT; j++) {
                const u32 i = wbase + j * 64;
                if (i < m) {
                    const u32 bin = (u32)(key[j] >> bshift) & (u32)(NBIN - 1);
                    const u32 s = s_cnt[bin], e = s_cnt[bin + 1];
                    u32 r = s;
                    for (u32 t = s; t < e; t++) {
                        const u64 k2 = grp[t];
                        const bool first = PAIRS ? (u32)s_idx[t] < i : t < pos[j];
                        r += (k2 < key[j] || (k2 == key[j] && first)) ? 1u : 0u;
                    }
                    pos[j] = r;
                }
            }
            __syncthreads();  // every walk is over: the final order goes into the same buffer
#pragma unroll
            for (int j = 0; j < LS_KPT; j++) {
                if (wbase + j * 64 < m) {
                    grp[pos[j]] = key[j];
                    if (PAIRS) s_vbuf[0][pos[j]] = val[j];
                }
            }
            __syncthreads();
            for (u32 i = (u32)tid; i < m; i += LS_NT) {
                a.keys[beg + i] = grp[i];
                if (PAIRS) a.vals[beg + i] = s_vbuf[0][i];
            }
            return;
        }
    }
    const u32 lt_lo = lane < 32 ? ((1u << lane) - 1u) : ~0u;
    const u32 lt_hi = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
    const int npass = (a.low_bits + RB - 1) / RB;
    int cur = 0;
    for (int p = 0; p < npass; p++) {
        const int shift = RB * p;
        for (int i = tid; i < LS_NW * RADIX / 2; i += LS_NT) reinterpret_cast<u32 *>(&s_wh[0][0])[i] = 0;
        __syncthreads();
        u32 rk[LS_KPT];
#pragma unroll
        for (int j = 0; j < LS_KPT; j++) {
            const u32 d = (u32)(key[j] >> shift) & DMASK;
            u32 plo, phi;
            match_any8(d, plo, phi);
            const u32 pre = s_wh[wave][d];
            rk[j] = pre + (u32)__popc(plo & lt_lo) + (u32)__popc(phi & lt_hi);
            s_wh[wave][d] = (unsigned short)(pre + (u32)__popc(plo) + (u32)__popc(phi));
        }
        __syncthreads();
        u32 cnt = 0;
#pragma unroll
        for (int w = 0; w < LS_NW; w++) {  // thread d: exclusive scan of digit d over the waves
            const u32 cw = s_wh[w][tid];
            s_wh[w][tid] = (unsigned short)cnt;
            cnt += cw;
        }
        u32 total = 0;
        const u32 dex = block_excl_scan_u32<LS_NT>(cnt, s_scan, &total);
        s_dex[tid] = dex;
        __syncthreads();
        u64 *dst = s_buf[cur];
        u32 *vdst = s_vbuf[PAIRS ? cur : 0];
#pragma unroll
        for (int j = 0; j < LS_KPT; j++) {
            const u32 d = (u32)(key[j] >> shift) & DMASK;
            const u32 pos = s_dex[d] + s_wh[wave][d] + rk[j];
            dst[pos] = key[j];
            if (PAIRS) vdst[pos] = val[j];
        }
        __syncthreads();
        if (p + 1 < npass) {
#pragma unroll
            for (int j = 0; j < LS_KPT; j++) {
                key[j] = dst[wbase + j * 64];
                if (PAIRS) val[j] = vdst[wbase + j * 64];
            }
        }
        cur ^= 1;  // (the next pass writes the other buffer: this one is still being read)
    }
    const u64 *res = s_buf[cur ^ 1];
    const u32 *vres = s_vbuf[PAIRS ? (cur ^ 1) : 0];
    for (u32 i = (u32)tid; i < m; i += LS_NT) {
        a.keys[beg + i] = res[i];
        if (PAIRS) a.vals[beg + i] = vres[i];
    }
}

bool sort_local_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_SORT_LOCAL", '0'); }  // developer knob: 0 = all passes through HBM

// 2^23 <= n < 2^32.  *done = false: not this route (narrow keys, too many oversized buckets): the caller runs the
// general passes over the keys as they are now (a permutation of the input).
// The buckets are the top `topb` bits with 2^topb ~ n / 1400: up to 1.3e8 keys the array is sorted by its top 16 bits (two
// scatter passes), beyond that by its top 24 (three passes, the result sits in the scratch copy and the bucket kernels
// write the caller's array from there).
// topb / npass of the bucket route for n keys (shared with ukm_sort_first_shift below)
static void ls_plan(u64 n, int *topb_out, int *npass_out) {
    int topb;
    if ((n >> 16) <= 2048) {
        topb = LS_TOP_MIN;
        while (topb < 16 && (n >> topb) > 1400) topb++;
    } else {
        topb = 17;
        while (topb < LS_TOP_MAX && (n >> topb) > 1400) topb++;
    }
    *topb_out = topb;
    *npass_out = topb <= 16 ? 2 : 3;
}

// first_hist (may be null): the 256-bin histogram of digit (key >> first_shift) & 255 over exactly these n keys, counted by
// whoever PRODUCED them (ukm_count: the encode kernel) -- it replaces this route's histogram pre-pass when the shift is the
// one the route would use
int sort_top16_local(ukm_ctx *c, u64 *keys, u32 *vals, u64 n, int key_bits, bool *done, const u64 *first_hist, int first_shift) {
    *done = false;
    // two passes as long as 65,536 buckets hold the keys (a third pass costs more than larger buckets)
    int topb, npass;
    ls_plan(n, &topb, &npass);
    const u32 nbuckets = 1u << topb;
    u64 *fh = nullptr, *gb = nullptr, *tk = nullptr, *start = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)MAX_PASSES * RADIX + 1, &fh));
    UKM_TRY(ws_alloc_t(c, (size_t)MAX_PASSES * RADIX, &gb));
    UKM_TRY(ws_alloc_t(c, (size_t)nbuckets + 2, &start));
    // (a context that has met crowded keys looks at a sample before anything else is spent on this route)
    bool heavy_seen = false;
    auto guard = [&](int low_bits) -> int {
        const u32 nsamp = 1u << 20;
        u32 *scnt = nullptr;
        u64 *sout = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)nbuckets, &scnt));
        UKM_TRY(ws_alloc_t(c, 2, &sout));
        UKM_HIP(hipMemsetAsync(scnt, 0, sizeof(u32) * (size_t)nbuckets, c->stream));
        UKM_HIP(hipMemsetAsync(sout, 0, 2 * sizeof(u64), c->stream));
        hipLaunchKernelGGL(ls_sample_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, keys, n, low_bits, topb, scnt, nsamp);
        // a bucket of 4096 keys holds 4096 * nsamp / n samples on average; 25 % above that to let borderline buckets pass
        const u32 thr = (u32)((double)(LS_NT * LS_CLASS_KPT[LS_NCLASS - 1]) * 1.25 * (double)nsamp / (double)n) + 4;
        hipLaunchKernelGGL(ls_sample_count_kernel, dim3(nbuckets / 256), dim3(256), 0, c->stream, scnt, thr, sout);
        UKM_HIP(hipGetLastError());
        u64 heavy[2] = {0, 0};
        UKM_TRY(ukm_read_u64(c, sout, heavy, 2));
        if (ukm_env(c, "UKM_SORT_DEBUG"))
            fprintf(stderr, "[sort] sample: %llu buckets look heavier than %u samples, %llu of %u samples in them\n", (unsigned long long)heavy[0], thr,
                    (unsigned long long)heavy[1], nsamp);
        heavy_seen = heavy[0] > (u64)LS_MAX_BIG || heavy[1] > (u64)nsamp / 4;
        return UKM_OK;
    };
    const int min_bits = 8 * npass + 16;  // (at least two digits for the buckets' own passes)
    if (c->sort_skew_seen && key_bits < 64) {
        if (key_bits < min_bits) return UKM_OK;
        UKM_TRY(guard(key_bits - topb));
        if (heavy_seen) return UKM_OK;
    }
    const unsigned hb = (unsigned)std::min<u64>((n + 4095) / 4096, (u64)c->num_cu * 8);
    int kb = key_bits;
    for (int attempt = 0; attempt < 2; attempt++) {
        UKM_HIP(hipMemsetAsync(fh, 0, (MAX_PASSES * RADIX + 1) * sizeof(u64), c->stream));
        u64 *or_all = (key_bits == 64 && attempt == 0) ? fh + (size_t)MAX_PASSES * RADIX : nullptr;
        if (first_hist && !or_all && first_shift == kb - 8 * npass) {
            UKM_HIP(hipMemcpyAsync(fh, first_hist, RADIX * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            c->stat_sort_fused_hist++;
        } else
            hipLaunchKernelGGL(radix_hist_kernel, dim3(hb), dim3(NT), 0, c->stream, keys, n, 1, fh, or_all, kb - 8 * npass);
        UKM_HIP(hipGetLastError());
        if (!or_all) break;
        u64 orv = 0;
        UKM_TRY(ukm_read_u64(c, or_all, &orv));  // (a caller that could not narrow key_bits: one read-back, as in the general route)
        const int bits = orv ? 64 - __builtin_clzll(orv) : 1;
        if (bits == 64) break;  // the histogram above is the right one
        // (57 .. 63 significant bits too: with kb left at 64 only 2^(bits - 48) of the top buckets would be populated, the
        //  route would run its scatter passes, give up and mark the context as one that has met crowded keys)
        kb = bits;
        if (kb < min_bits) return UKM_OK;
    }
    if (kb < min_bits) return UKM_OK;  // (narrow keys)
    const int low_bits = kb - topb;
    if (c->sort_skew_seen && key_bits == 64) UKM_TRY(guard(low_bits));
    if (heavy_seen) return UKM_OK;
    int sh[3];
    for (int i = 0; i < npass; i++) sh[i] = kb - 8 * (npass - i);
    u32 *tv = nullptr;
    UKM_TRY(ws_alloc_t(c, n, &tk));
    if (vals) UKM_TRY(ws_alloc_t(c, n, &tv));
    bool in_tmp = false;
    if (vals) {
        if (n < (1ull << 30)) UKM_TRY((run_passes<u32, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, npass, sh, gb, &in_tmp, fh)));
        else UKM_TRY((run_passes<u64, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, npass, sh, gb, &in_tmp, fh)));
    } else {
        if (n < (1ull << 30)) UKM_TRY((run_passes<u32, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, nullptr, tk, nullptr, n, npass, sh, gb, &in_tmp, fh)));
        else UKM_TRY((run_passes<u64, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, nullptr, tk, nullptr, n, npass, sh, gb, &in_tmp, fh)));
    }
    const u64 *src = in_tmp ? tk : keys;   // (three passes end in the scratch copy)
    const u32 *vsrc = in_tmp ? tv : vals;
    auto give_up = [&]() -> int {          // the general passes work on the caller's array
        if (in_tmp) {
            UKM_HIP(hipMemcpyAsync(keys, tk, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            if (vals) UKM_HIP(hipMemcpyAsync(vals, tv, n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        }
        return UKM_OK;
    };
    hipLaunchKernelGGL(ls_bounds_kernel, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, c->stream, src, n, low_bits, topb, start);
    // Buckets differ in size (canonical k-mers: twice the average at the low end of the code space, none at the top), so
    // every bucket goes to the instantiation of ITS size class: a tiny kernel lists the buckets of every class, one launch
    // per class that occurs, and the few buckets beyond 4096 keys are sorted by the general route.
    u64 *cls = nullptr;
    u32 *ids = nullptr;
    static_assert(LS_NCLASS + 3 <= 64, "read-back through the scratch");
    if (!c->sort_stat_dev) {
        UKM_HIP(hipMalloc((void **)&c->sort_stat_dev, 64));
        UKM_HIP(hipMemsetAsync(c->sort_stat_dev, 0, 64, c->stream));
    }
    u32 *big_ids = nullptr;
    u64 *big_size = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)LS_NCLASS + 3, &cls));
    UKM_TRY(ws_alloc_t(c, (size_t)LS_NCLASS << topb, &ids));
    UKM_TRY(ws_alloc_t(c, (size_t)LS_MAX_BIG, &big_ids));
    UKM_TRY(ws_alloc_t(c, (size_t)nbuckets + 1, &big_size));
    UKM_HIP(hipMemsetAsync(cls, 0, (LS_NCLASS + 3) * sizeof(u64), c->stream));
    hipLaunchKernelGGL(ls_classify_kernel, dim3(nbuckets / 256), dim3(256), 0, c->stream, start, cls, ids, topb, big_ids, big_size, c->sort_stat_dev);
    UKM_HIP(hipGetLastError());
    u64 hc[LS_NCLASS + 3];
    UKM_TRY(ukm_read_u64(c, cls, hc, LS_NCLASS + 3));
    // (hc[LS_NCLASS + 2]: buckets of the PREVIOUS bucket-route sort on this context that fell back to the digit passes.  More
    //  than half of them: keys with several copies each, the next 15 sorts do not try the counting step)
    if (c->sort_last_counting && hc[LS_NCLASS + 2] * 2 > c->sort_last_buckets) c->sort_counting_skip = 15;
    const u64 big_keys = hc[LS_NCLASS + 1];
    // oversized buckets are gathered and sorted by ONE call of the general route: worth it for a few of them holding a
    // minor share of the keys, else the general passes sort everything (keys crowded into few buckets)
    if (hc[LS_NCLASS] > (u64)LS_MAX_BIG || big_keys > n / 4) {
        c->sort_skew_seen = true;        // (and the next sort on this context looks at a sample before it tries)
        if (ukm_env(c, "UKM_SORT_DEBUG")) fprintf(stderr, "[sort] %llu buckets beyond every class: general route\n", (unsigned long long)hc[LS_NCLASS]);
        return give_up();
    }
    LocalSortArgs a;
    a.src = src; a.vsrc = vsrc; a.keys = keys; a.vals = vals; a.n = n; a.start = start; a.low_bits = low_bits;
    a.counting = ukm_env_is(c, "UKM_SORT_COUNTING", '0') ? 0 : 1;  // developer knob: 0 = digit passes in every bucket
    if (a.counting && c->sort_counting_skip > 0 && !ukm_env_is(c, "UKM_SORT_COUNTING", '1')) {
        a.counting = 0;
        c->sort_counting_skip--;
    }
    a.stat = c->sort_stat_dev;
    c->sort_last_counting = a.counting != 0;
    c->sort_last_buckets = 0;
    for (int k = 0; k < LS_NCLASS; k++) c->sort_last_buckets += hc[k];
    const dim3 block(LS_NT);
    // The classes' launches do not depend on each other and the small ones leave most of the chip idle (canonical k-mers:
    // ten classes occur, eight of them 16-60 us each): they go round robin over the call's stream and two side streams,
    // largest class first, forked from / joined to the call's stream by events.
    int nocc = 0;
    for (int k = 0; k < LS_NCLASS; k++) nocc += hc[k] != 0;
    const bool fan = nocc >= 3 && !ukm_env_is(c, "UKM_SORT_FAN", '0');
    if (fan) {
        if (!c->ev_fork) {
            UKM_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            for (int i = 0; i < 2; i++) {
                UKM_HIP(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
                UKM_HIP(hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming));
            }
        }
        UKM_HIP(hipEventRecord(c->ev_fork, c->stream));
        for (int i = 0; i < 2; i++) UKM_HIP(hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
    }
    int turn = 0;
    hipStream_t lst = c->stream;
#define LS_LAUNCH(K)                                                                                          \
    do {                                                                                                      \
        if (vals) hipLaunchKernelGGL((ls_sort_kernel<K, true>), grid, block, 0, lst, a);                      \
        else hipLaunchKernelGGL((ls_sort_kernel<K, false>), grid, block, 0, lst, a);                          \
    } while (0)
    for (int kk = 0; kk < LS_NCLASS; kk++) {
        const int k = LS_NCLASS - 1 - kk;  // (the classes with the most keys per bucket first)
        if (hc[k] == 0) continue;
        if (fan) lst = turn % 3 == 0 ? c->stream : c->side[turn % 3 - 1];
        turn++;
        a.ids = ids + ((size_t)k << topb);
        const dim3 grid((unsigned)hc[k]);
        switch (LS_CLASS_KPT[k]) {
        case 1: LS_LAUNCH(1); break;
        case 2: LS_LAUNCH(2); break;
        case 3: LS_LAUNCH(3); break;
        case 4: LS_LAUNCH(4); break;
        case 5: LS_LAUNCH(5); break;
        case 6: LS_LAUNCH(6); break;
        case 7: LS_LAUNCH(7); break;
        case 8: LS_LAUNCH(8); break;
        case 10: LS_LAUNCH(10); break;
        case 12: LS_LAUNCH(12); break;
        case 14: LS_LAUNCH(14); break;
        default: LS_LAUNCH(16); break;
        }
    }
    UKM_HIP(hipGetLastError());
    if (fan)
        for (int i = 0; i < 2; i++) {
            UKM_HIP(hipEventRecord(c->ev_side[i], c->side[i]));
            UKM_HIP(hipStreamWaitEvent(c->stream, c->ev_side[i], 0));
        }
    if (hc[LS_NCLASS]) {
        // The oversized buckets -- low-complexity k-mers of a real genome crowd a few dozen to a few thousand of them --
        // side by side in a scratch array, in bucket order: sorted by the whole key they stay side by side (their top bits
        // differ), each one sorted; one general sort instead of one per bucket (48 small sorts took 8 ms).  Where a bucket
        // lies in that array is the exclusive scan of the oversized buckets' sizes over the bucket index, so gather and
        // scatter are one kernel each, whatever the number of buckets (round 3 read the list back and issued two copies
        // per bucket from the host, which is why it gave up beyond 32 of them).
        const unsigned nb = (unsigned)hc[LS_NCLASS];
        u64 *big_off = nullptr, *bk = nullptr, *tot = nullptr;
        u32 *bv = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)nbuckets + 1, &big_off));
        UKM_TRY(ws_alloc_t(c, 1, &tot));
        UKM_TRY(ws_alloc_t(c, big_keys, &bk));
        if (vals) UKM_TRY(ws_alloc_t(c, big_keys, &bv));
        UKM_TRY(ukm_dev_exclusive_scan_u64(c, big_size, big_off, nbuckets, tot));
        const dim3 cg(nb, 16);
        hipLaunchKernelGGL(ls_big_copy_kernel<false>, cg, dim3(256), 0, c->stream, big_ids, start, big_off, src, vsrc, bk, bv);
        UKM_HIP(hipGetLastError());
        c->sort_general_only = true;  // (crowded by construction: not through this route again)
        const int src_rc = ukm_dev_sort(c, bk, bv, big_keys, kb);
        c->sort_general_only = false;
        UKM_TRY(src_rc);
        hipLaunchKernelGGL(ls_big_copy_kernel<true>, cg, dim3(256), 0, c->stream, big_ids, start, big_off, bk, bv, keys, vals);
        UKM_HIP(hipGetLastError());
    }
#undef LS_LAUNCH
    *done = true;
    return UKM_OK;
}

}  // namespace

// keys (and vals, may be NULL) are device pointers; sorted in place (stable for pairs)
// The shift of the digit whose histogram the bucket route's pre-pass would build for n keys of key_bits bits, or -1 when
// a sort of such keys does not take that route (or reads an OR of all keys first: key_bits = 64).
int ukm_sort_first_shift(const ukm_ctx *c, u64 n, int key_bits) {
    if (key_bits <= 0 || key_bits >= 64 || n >= (1ull << 32) || n < (1ull << 23) || RB != 8 || key_bits < 32) return -1;
    if (!sort_local_enabled(c) || c->sort_general_only) return -1;
    int topb, npass;
    ls_plan(n, &topb, &npass);
    if (key_bits < 8 * npass + 16) return -1;
    return key_bits - 8 * npass;
}

int ukm_dev_sort(ukm_ctx *c, u64 *keys, u32 *vals, u64 n, int key_bits) { return ukm_dev_sort_hist(c, keys, vals, n, key_bits, nullptr, -1); }

int ukm_dev_sort_hist(ukm_ctx *c, u64 *keys, u32 *vals, u64 n, int key_bits, const u64 *first_hist, int first_shift) {
    if (n < 2) return UKM_OK;
    if (key_bits <= 0 || key_bits > 64) key_bits = 64;
    if (n >= (1ull << 32)) {
        // The onesweep tile / status arithmetic is 32-bit: larger inputs are sorted as chunks of 2^31 records
        // and combined by the keep-everything 2-way merge (pairwise tree, ping-pong between the input and a
        // scratch copy) -- the reference's `sort -m` protocol, in HBM.
        const u64 CH = 1ull << 31;
        const u64 nch = (n + CH - 1) / CH;
        for (u64 i = 0; i < nch; i++) {
            const u64 m = std::min<u64>(CH, n - i * CH);
            WsMark mark = ws_mark(c);
            UKM_TRY(ukm_dev_sort(c, keys + i * CH, vals ? vals + i * CH : nullptr, m, key_bits));
            ws_release(c, mark);
        }
        u64 *tk = nullptr;
        u32 *tv = nullptr;
        UKM_TRY(ws_alloc_t(c, n, &tk));
        if (vals) UKM_TRY(ws_alloc_t(c, n, &tv));
        u64 *src_k = keys, *dst_k = tk;
        u32 *src_v = vals, *dst_v = tv;
        for (u64 run = CH; run < n; run *= 2) {  // runs of `run` records are sorted; merge neighbours
            for (u64 lo = 0; lo < n; lo += 2 * run) {
                const u64 na = std::min<u64>(run, n - lo);
                const u64 nb = (lo + run < n) ? std::min<u64>(run, n - lo - run) : 0;
                if (nb == 0) {
                    UKM_HIP(hipMemcpyAsync(dst_k + lo, src_k + lo, na * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
                    if (vals) UKM_HIP(hipMemcpyAsync(dst_v + lo, src_v + lo, na * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
                    continue;
                }
                u64 nm = 0;
                WsMark mark = ws_mark(c);
                UKM_TRY(ukm_dev_setop2(c, UKM_OP_MERGE_INTERNAL, src_k + lo, vals ? src_v + lo : nullptr, na, src_k + lo + run,
                                       vals ? src_v + lo + run : nullptr, nb, 0, dst_k + lo, vals ? dst_v + lo : nullptr,
                                       na + nb, &nm));
                ws_release(c, mark);
            }
            std::swap(src_k, dst_k);
            std::swap(src_v, dst_v);
        }
        if (src_k != keys) {
            UKM_HIP(hipMemcpyAsync(keys, src_k, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            if (vals) UKM_HIP(hipMemcpyAsync(vals, src_v, n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        }
        return UKM_OK;
    }
    const int passes = (key_bits + RB - 1) / RB;
#ifndef SORT_FUSED_MIN
#define SORT_FUSED_MIN (1ull << 24)
#endif
#ifndef SORT_LOCAL_MIN
#define SORT_LOCAL_MIN (1ull << 23)  // (measured: 1.2e7 keys 0.62 -> 0.50 ms, 6.7e6 equal, 1.5e6 slower)
#endif
    if (n >= SORT_LOCAL_MIN && RB == 8 && key_bits >= 32 && sort_local_enabled(c) && !c->sort_general_only) {
        // two passes over the top 16 bits, then every bucket in LDS (above); stable, so taxids may ride along
        WsMark mark = ws_mark(c);
        bool done = false;
        const int rc = sort_top16_local(c, keys, vals, n, key_bits, &done, first_hist, first_shift);
        ws_release(c, mark);
        UKM_TRY(rc);
        if (done) return UKM_OK;
    }
    if (n >= SORT_FUSED_MIN) {
        // Large inputs: only the FIRST digit's histogram is built by a pass over the keys; every scatter pass counts
        // the next digit on the fly and a 256-thread kernel turns the counts into bases between two passes.  The
        // histograms never leave the device and the passes take their tile ids from a ticket counter (SORT_TICKET,
        // no watchdog flag to read back), so a sort with a known key width has no host round trip at all; constant
        // digits inside the key width are not detected here.
        u64 *fh = nullptr, *gb = nullptr, *tk = nullptr;
        u32 *tv = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)MAX_PASSES * RADIX + 1, &fh));
        UKM_TRY(ws_alloc_t(c, (size_t)MAX_PASSES * RADIX, &gb));
        UKM_HIP(hipMemsetAsync(fh, 0, (MAX_PASSES * RADIX + 1) * sizeof(u64), c->stream));
        unsigned hb = (unsigned)std::min<u64>((n + 4095) / 4096, (u64)c->num_cu * 8);
        // A caller that could not narrow key_bits (64) gets ONE 8-byte read-back: the OR of all keys, gathered by the
        // pre-pass for free, drops the passes over high digits that are zero everywhere (hashes restricted by a
        // Scaled threshold, k <= 28 codes handed over as plain uint64: 5-7 passes instead of 8).
        u64 *or_all = key_bits == 64 ? fh + (size_t)MAX_PASSES * RADIX : nullptr;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(hb), dim3(NT), 0, c->stream, keys, n, 1, fh, or_all);
        UKM_HIP(hipGetLastError());
        int passes_f = passes;
        if (or_all) {
            u64 orv = 0;
            UKM_TRY(ukm_read_u64(c, or_all, &orv));
            const int bits = orv ? 64 - __builtin_clzll(orv) : 1;
            passes_f = (bits + RB - 1) / RB;
        }
        int sh[MAX_PASSES];
        for (int p = 0; p < passes_f; p++) sh[p] = RB * p;
        UKM_TRY(ws_alloc_t(c, n, &tk));
        if (vals) UKM_TRY(ws_alloc_t(c, n, &tv));
        bool in_tmp = false;
        if (vals) {
            if (n < (1ull << 30)) UKM_TRY((run_passes<u32, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, passes_f, sh, gb, &in_tmp, fh)));
            else UKM_TRY((run_passes<u64, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, passes_f, sh, gb, &in_tmp, fh)));
        } else {
            if (n < (1ull << 30)) UKM_TRY((run_passes<u32, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, vals, tk, tv, n, passes_f, sh, gb, &in_tmp, fh)));
            else UKM_TRY((run_passes<u64, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, vals, tk, tv, n, passes_f, sh, gb, &in_tmp, fh)));
        }
        if (in_tmp) {
            UKM_HIP(hipMemcpyAsync(keys, tk, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            if (vals) UKM_HIP(hipMemcpyAsync(vals, tv, n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        }
        return UKM_OK;
    }

    u64 *ghist = nullptr;
    UKM_TRY(ws_alloc_t(c, MAX_PASSES * RADIX, &ghist));
    UKM_HIP(hipMemsetAsync(ghist, 0, MAX_PASSES * RADIX * sizeof(u64), c->stream));
    unsigned hblocks = (unsigned)std::min<u64>((n + 4095) / 4096, (u64)c->num_cu * 8);
    hipLaunchKernelGGL(radix_hist_kernel, dim3(hblocks), dim3(NT), 0, c->stream, keys, n, passes, ghist, (u64 *)nullptr);
    UKM_HIP(hipGetLastError());
    std::vector<u64> h((size_t)passes * RADIX);
    UKM_HIP(hipMemcpyAsync(h.data(), ghist, h.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));

    int shifts[MAX_PASSES];
    std::vector<u64> gb;
    int npass = 0;
    for (int p = 0; p < passes; p++) {
        const u64 *hp = &h[(size_t)p * RADIX];
        bool constant = false;
        for (int d = 0; d < RADIX; d++)
            if (hp[d] == n) constant = true;
        if (constant) continue;  // every key has the same digit: the pass is the identity
        shifts[npass++] = RB * p;
        u64 sum = 0;
        for (int d = 0; d < RADIX; d++) {
            gb.push_back(sum);
            sum += hp[d];
        }
    }
    if (npass == 0) return UKM_OK;

    u64 *gbase_dev = nullptr, *tk = nullptr;
    u32 *tv = nullptr;
    UKM_TRY(ws_alloc_t(c, gb.size(), &gbase_dev));
    UKM_HIP(hipMemcpyAsync(gbase_dev, gb.data(), gb.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // gb is a local
    UKM_TRY(ws_alloc_t(c, n, &tk));
    if (vals) UKM_TRY(ws_alloc_t(c, n, &tv));
    bool in_tmp = false;
    if (vals) {
        if (n < (1ull << 30)) UKM_TRY((run_passes<u32, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, npass, shifts, gbase_dev, &in_tmp)));
        else UKM_TRY((run_passes<u64, true, SORT_NT_PAIRS, SORT_VT_PAIRS>(c, keys, vals, tk, tv, n, npass, shifts, gbase_dev, &in_tmp)));
    } else {
        if (n < (1ull << 30)) UKM_TRY((run_passes<u32, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, vals, tk, tv, n, npass, shifts, gbase_dev, &in_tmp)));
        else UKM_TRY((run_passes<u64, false, SORT_NT_KEYS, SORT_VT_KEYS>(c, keys, vals, tk, tv, n, npass, shifts, gbase_dev, &in_tmp)));
    }
    if (in_tmp) {
        UKM_HIP(hipMemcpyAsync(keys, tk, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        if (vals) UKM_HIP(hipMemcpyAsync(vals, tv, n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    }
    return UKM_OK;
}

extern "C" int ukm_sort_u64(ukm_ctx *ctx, uint64_t *keys, uint64_t n, int key_bits) {
    if (!ctx || (!keys && n)) UKM_FAIL(UKM_ERR_INVALID, "ukm_sort_u64: NULL argument");
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        void *d = nullptr;
        UKM_TRY(ukm_inout(ctx, keys, n * sizeof(u64), &d));
        return ukm_dev_sort(ctx, (u64 *)d, nullptr, n, key_bits);
    }();
    return ukm_finish(&s, rc);
}

extern "C" int ukm_sort_pairs(ukm_ctx *ctx, uint64_t *keys, uint32_t *taxids, uint64_t n, int key_bits) {
    if (!ctx || (!keys && n) || (!taxids && n)) UKM_FAIL(UKM_ERR_INVALID, "ukm_sort_pairs: NULL argument");
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        void *dk = nullptr, *dv = nullptr;
        UKM_TRY(ukm_inout(ctx, keys, n * sizeof(u64), &dk));
        UKM_TRY(ukm_inout(ctx, taxids, n * sizeof(u32), &dv));
        return ukm_dev_sort(ctx, (u64 *)dk, (u32 *)dv, n, key_bits);
    }();
    return ukm_finish(&s, rc);
}
