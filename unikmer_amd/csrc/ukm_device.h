// ukm_device.h — device-side building blocks for gfx950 (wave64): wave/block scans,
// single-pass decoupled look-back over tile aggregates, device LCA.
#pragma once

#include "ukm_internal.h"

#define UKM_WAVE 64

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- wave64 scans ---------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
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
#define LB_AGG (1ull << 62)
#define LB_INCL (2ull << 62)
#define LB_VAL ((1ull << 62) - 1)

__device__ __forceinline__ void lb_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ONE full wave of the block (all 64 lanes).  Publishes this tile's aggregate, walks
// predecessors 64 at a time, publishes the inclusive prefix, returns the exclusive prefix
// (same value in every lane).  Forward progress: tile ids are handed out by an atomic ticket,
// so every predecessor tile is already running.
__device__ __forceinline__ u64 lb_lookback(u64 *status, u64 tile, u64 agg) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) lb_store(&status[0], LB_INCL | agg);
        return 0;
    }
    if (lane == 0) lb_store(&status[tile], LB_AGG | agg);
    u64 excl = 0;
    long long base = (long long)tile - 1;
    for (;;) {
        long long idx = base - lane;
        u64 w = (idx >= 0) ? lb_load(&status[idx]) : LB_INCL;
        u64 st = w >> 62;
        u64 incl_mask = __ballot(st == 2);
        u64 empty_mask = __ballot(st == 0);
        int first_incl = incl_mask ? __builtin_ctzll(incl_mask) : 64;
        u64 need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1);
        if (empty_mask & need) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        u64 v = (lane <= first_incl) ? (w & LB_VAL) : 0;
        excl += wave_reduce_sum_u64(v);
        if (first_incl < 64) break;
        base -= 64;
    }
    if (lane == 0) lb_store(&status[tile], LB_INCL | (excl + agg));
    return excl;
}

// ---- device LCA (contract: include/unikmer_hip.h, ukm_taxonomy_load) ---------------------------
__device__ __forceinline__ u32 tax_resolve(const TaxDev &T, u32 a) {
    if (a >= T.size) return 0;
    if (T.parent[a]) return a;
    if (T.merged) {
        u32 b = T.merged[a];
        if (b && b < T.size && T.parent[b]) return b;
    }
    return 0;
}

__device__ __forceinline__ u32 lca_dev(const TaxDev &T, u32 a, u32 b) {
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    a = tax_resolve(T, a);
    b = tax_resolve(T, b);
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    int da = T.depth[a], db = T.depth[b];
    while (da > db) { a = T.parent[a]; da--; }
    while (db > da) { b = T.parent[b]; db--; }
    while (a != b) {
        if (da == 0) return 0;  // different trees
        a = T.parent[a];
        b = T.parent[b];
        da--;
    }
    return a;
}
