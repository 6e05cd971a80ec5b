// ukm_scan.hip — scans over a SORTED (code [, taxid]) stream: the replacement for the linear
// dedup / LCA-fold / "repeated" loops of sort.go:484-572 and dumpCodes[Taxids]2File
// (util-sort.go:35-190).  Plus two small primitives used across the library: sortedness check
// and a single-pass exclusive scan of uint64.
//
// Unique kernel: one 256-thread workgroup per tile of NT*VT records; keys are loaded coalesced
// into LDS with a one-element halo on both sides; run heads are found by adjacent comparison;
// each head emits 0, 1 or 2 records according to the mode; the per-code taxid is the LCA fold
// over the run (associative + commutative on a tree, so fold order is immaterial); output
// offsets come from the same single-pass decoupled look-back as the set operations, so the
// stream is read once and written once: 8n (+4n) bytes in, 8u (+4u) bytes out.
#include <algorithm>

#include "ukm_device.h"

#define UKM_UNIQUE_LAST 5  // internal: one record per run, the LAST record of the run
#define UKM_COMMON 6       // internal: run heads whose run length >= threshold (common.go:331-335)

namespace {

constexpr int NT = 256;
constexpr int VT = 8;
constexpr int TILE = NT * VT;

struct UniqArgs {
    const u64 *k;
    const u32 *t;
    u64 n;
    u64 *out;
    u32 *tout;
    u64 out_cap;
    u64 *status;
    u32 *ticket;
    u64 *result;  // [0] total, [1] flags (bit1 = unsorted)
    u64 ntiles;
    TaxDev tax;
    int mode;
    u32 threshold;
};

template <bool TAX>
__global__ __launch_bounds__(NT) void unique_tile_kernel(UniqArgs p) {
    __shared__ u64 s_keys[TILE + 2];
    __shared__ u64 s_outk[2 * TILE];  // REPEATED_CHUNK can emit two records per input
    __shared__ u32 s_outt[TAX ? 2 * TILE : 1];
    __shared__ u32 s_scan[NT / 64 + 1];
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x;
    if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const u64 tile = s_misc[0];
    const u64 i0 = tile * (u64)TILE;
    const int cnt_t = (int)((p.n - i0 < (u64)TILE) ? (p.n - i0) : (u64)TILE);

    // slot j holds k[i0 + j - 1]
    for (int j = tid; j < cnt_t + 2; j += NT) {
        long long g = (long long)i0 + j - 1;
        s_keys[j] = (g >= 0 && (u64)g < p.n) ? p.k[g] : 0;
    }
    __syncthreads();

    u64 ok[2 * VT];
    u32 ot[2 * VT];
    int ne = 0;
    u32 bad = 0;
    const int mode = p.mode;
#pragma unroll
    for (int s = 0; s < VT; s++) {
        const int j = tid * VT + s;  // local index
        if (j < cnt_t) {
            const u64 gi = i0 + j;
            const u64 key = s_keys[j + 1];
            const bool has_prev = gi > 0, has_next = gi + 1 < p.n;
            const u64 prev = s_keys[j], next = s_keys[j + 2];
            if (has_prev && prev > key) bad |= 2;
            const bool head = !has_prev || prev != key;
            const bool tail = !has_next || next != key;
            int emit = 0;
            u32 tx = 0;
            if (mode == UKM_UNIQUE_LAST) {
                emit = tail ? 1 : 0;
                if (TAX && tail) tx = p.t[gi];
            } else if (head && mode == UKM_COMMON) {
                u64 len = 1;
                for (u64 q = gi + 1; q < p.n && p.k[q] == key && (TAX || len < p.threshold); q++) len++;
                emit = (len >= p.threshold) ? 1 : 0;
                if (TAX && emit) {
                    tx = p.t[gi];
                    for (u64 q = gi + 1; q < gi + len; q++) tx = lca_dev(p.tax, tx, p.t[q]);  // common.go:265
                }
            } else if (head) {
                const bool repeated = !tail;
                if (mode == UKM_UNIQUE) emit = 1;
                else if (mode == UKM_REPEATED) emit = repeated ? 1 : 0;
                else if (mode == UKM_SINGLETON) emit = repeated ? 0 : 1;
                else emit = repeated ? 2 : 1;  // UKM_REPEATED_CHUNK
                if (TAX && emit) {
                    tx = p.t[gi];
                    for (u64 q = gi + 1; q < p.n && p.k[q] == key; q++)
                        tx = lca_dev(p.tax, p.t[q], tx);  // sort.go:491 lca = LCA(taxid, lca)
                }
            }
            if (emit >= 1) { ok[ne] = key; ot[ne] = tx; ne++; }
            if (emit == 2) { ok[ne] = key; ot[ne] = tx; ne++; }
        }
    }
    u32 tile_total;
    const u32 excl = block_excl_scan_u32<NT>((u32)ne, s_scan, &tile_total);
#pragma unroll
    for (int s = 0; s < 2 * VT; s++)
        if (s < ne) {
            s_outk[excl + s] = ok[s];
            if (TAX) s_outt[excl + s] = ot[s];
        }
    if (tid < 64) {
        u64 base = lb_lookback(p.status, tile, (u64)tile_total);
        if (tid == 0) s_misc[1] = base;
    }
    if (bad) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)bad);
    __syncthreads();
    const u64 base = s_misc[1];
    for (u32 i = (u32)tid; i < tile_total; i += NT) {
        u64 pos = base + i;
        if (pos < p.out_cap) {
            p.out[pos] = s_outk[i];
            if (TAX) p.tout[pos] = s_outt[i];
        }
    }
    if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
}

__global__ void check_sorted_kernel(const u64 *k, u64 n, u32 *flags) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 f = 0;
    for (; i + 1 < n; i += stride) {
        u64 a = k[i], b = k[i + 1];
        if (a > b) f |= 2;
        else if (a == b) f |= 1;
    }
    if (__any(f != 0)) {
        u32 m = f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
        if (lane_id() == 0) atomicOr(flags, m);
    }
}

// single-pass exclusive scan of u64, tile = 256 x 8
constexpr int SCAN_VT = 8;
__global__ __launch_bounds__(NT) void excl_scan_u64_kernel(const u64 *in, u64 *out, u64 n, u64 *status,
                                                         u32 *ticket, u64 *total_out, u64 ntiles) {
    __shared__ u64 s_w[NT / 64];
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (tid == 0) s_misc[0] = (u64)atomicAdd(ticket, 1u);
    __syncthreads();
    const u64 tile = s_misc[0];
    const u64 i0 = tile * (u64)(NT * SCAN_VT) + (u64)tid * SCAN_VT;
    u64 v[SCAN_VT];
    u64 sum = 0;
#pragma unroll
    for (int s = 0; s < SCAN_VT; s++) {
        v[s] = (i0 + s < n) ? in[i0 + s] : 0;
        sum += v[s];
    }
    // wave inclusive scan of sums
    u64 incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u64 wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        u64 t = s_w[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    if (tid < 64) {
        u64 base = lb_lookback(status, tile, tot);
        if (tid == 0) s_misc[1] = base;
    }
    __syncthreads();
    u64 run = s_misc[1] + wbase + incl - sum;
#pragma unroll
    for (int s = 0; s < SCAN_VT; s++) {
        if (i0 + s < n) out[i0 + s] = run;
        run += v[s];
    }
    if (tid == 0 && tile == ntiles - 1 && total_out) *total_out = s_misc[1] + tot;
}

}  // namespace

int ukm_dev_check_sorted(ukm_ctx *c, const u64 *keys, u64 n, bool *sorted, bool *strict) {
    *sorted = true;
    *strict = true;
    if (n < 2) return UKM_OK;
    u64 *flags = nullptr;
    UKM_TRY(ws_alloc_t(c, 1, &flags));
    UKM_HIP(hipMemsetAsync(flags, 0, sizeof(u64), c->stream));
    unsigned blocks = (unsigned)std::min<u64>((n + 255) / 256, (u64)c->num_cu * 16);
    hipLaunchKernelGGL(check_sorted_kernel, dim3(blocks), dim3(256), 0, c->stream, keys, n, (u32 *)flags);
    UKM_HIP(hipGetLastError());
    u64 f = 0;
    UKM_TRY(ukm_read_u64(c, flags, &f));
    *sorted = (f & 2) == 0;
    *strict = (f & 3) == 0;
    return UKM_OK;
}

int ukm_dev_exclusive_scan_u64(ukm_ctx *c, const u64 *in, u64 *out, u64 n, u64 *total_dev) {
    if (n == 0) {
        if (total_dev) UKM_HIP(hipMemsetAsync(total_dev, 0, sizeof(u64), c->stream));
        return UKM_OK;
    }
    const u64 ntiles = (n + NT * SCAN_VT - 1) / (NT * SCAN_VT);
    u64 *ctl = nullptr;
    const size_t nctl = 8 + lb_status_words(ntiles);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
    hipLaunchKernelGGL(excl_scan_u64_kernel, dim3((unsigned)ntiles), dim3(NT), 0, c->stream, in, out, n,
                       ctl + 8, (u32 *)ctl, total_dev, ntiles);
    UKM_HIP(hipGetLastError());
    return UKM_OK;
}

int ukm_dev_unique(ukm_ctx *c, const u64 *keys, const u32 *taxids, u64 n, int mode, u64 *out,
                   u32 *tout, u64 out_cap, u64 *n_out) {
    return ukm_dev_unique_ex(c, keys, taxids, n, mode, 0, out, tout, out_cap, n_out);
}

int ukm_dev_unique_ex(ukm_ctx *c, const u64 *keys, const u32 *taxids, u64 n, int mode, u32 threshold,
                      u64 *out, u32 *tout, u64 out_cap, u64 *n_out) {
    if (mode < UKM_PLAIN || mode > UKM_COMMON) UKM_FAIL(UKM_ERR_INVALID, "ukm_unique: unknown mode %d", mode);
    const bool tax = taxids != nullptr;
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "ukm_unique: taxids given but out_taxids is NULL");
    *n_out = 0;
    if (n == 0) return UKM_OK;
    if (mode == UKM_PLAIN) {  // sort.go:566-572: every record kept
        *n_out = n;
        if (n > out_cap) UKM_FAIL(UKM_ERR_CAPACITY, "ukm_unique: output needs %llu records", (unsigned long long)n);
        if (out != keys) UKM_HIP(hipMemcpyAsync(out, keys, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        if (tax && tout != taxids)
            UKM_HIP(hipMemcpyAsync(tout, taxids, n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        return UKM_OK;
    }
    if (tax && mode != UKM_UNIQUE_LAST && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_unique: records carry taxids but no taxonomy is loaded");
    UniqArgs p;
    memset(&p, 0, sizeof(p));
    p.k = keys; p.t = taxids; p.n = n;
    p.out = out; p.tout = tout; p.out_cap = out_cap;
    p.ntiles = (n + TILE - 1) / TILE;
    p.tax = ukm_taxdev(c);
    p.mode = mode;
    p.threshold = threshold;
    u64 *ctl = nullptr;
    const size_t nctl = 8 + lb_status_words(p.ntiles);
    UKM_TRY(ws_alloc_t(c, nctl, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
    p.result = ctl;
    p.ticket = (u32 *)(ctl + 2);
    p.status = ctl + 8;
    if (tax) hipLaunchKernelGGL(unique_tile_kernel<true>, dim3((unsigned)p.ntiles), dim3(NT), 0, c->stream, p);
    else hipLaunchKernelGGL(unique_tile_kernel<false>, dim3((unsigned)p.ntiles), dim3(NT), 0, c->stream, p);
    UKM_HIP(hipGetLastError());
    u64 res[2];
    UKM_TRY(ukm_read_u64(c, p.result, res, 2));
    if (res[1] & 2) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_unique: input stream is not sorted");
    *n_out = res[0];
    if (res[0] > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "ukm_unique: output needs %llu records, capacity is %llu",
                 (unsigned long long)res[0], (unsigned long long)out_cap);
    return UKM_OK;
}

extern "C" int ukm_unique(ukm_ctx *ctx, const uint64_t *keys, const uint32_t *taxids, uint64_t n,
                          int mode, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap,
                          uint64_t *n_out) {
    if (!ctx || !n_out || (!keys && n) || (!out_keys && out_cap))
        UKM_FAIL(UKM_ERR_INVALID, "ukm_unique: NULL argument");
    if (mode < UKM_PLAIN || mode > UKM_SINGLETON) UKM_FAIL(UKM_ERR_INVALID, "ukm_unique: unknown mode %d", mode);
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *k = nullptr;
        const u32 *t = nullptr;
        u64 *out = nullptr;
        u32 *tout = nullptr;
        UKM_TRY(ukm_in_t(ctx, keys, n, &k));
        UKM_TRY(ukm_in_t(ctx, taxids, n, &t));
        UKM_TRY(ukm_out_t(ctx, out_keys, out_cap, &out));
        UKM_TRY(ukm_out_t(ctx, out_taxids, out_cap, &tout));
        int r = ukm_dev_unique(ctx, k, t, n, mode, out, tout, out_cap, n_out);
        u64 m = (r == UKM_OK) ? *n_out : 0;
        ukm_out_resize(ctx, out_keys, m * sizeof(u64));
        if (out_taxids) ukm_out_resize(ctx, out_taxids, m * sizeof(u32));
        return r;
    }();
    return ukm_finish(&s, rc);
}
