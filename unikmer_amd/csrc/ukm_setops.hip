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

constexpr int NT = 256;

enum { FLAG_DUP = 1, FLAG_UNSORTED = 2 };

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
};

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

template <bool RANK>
__global__ void setop_partition_kernel(SetopArgs p, int tile_items) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > p.ntiles) return;
    const u64 N = p.na + p.nb;
    u64 diag = t * (u64)tile_items;
    if (diag > N) diag = N;
    u64 lo = diag > p.nb ? diag - p.nb : 0;
    u64 hi = diag < p.na ? diag : p.na;
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        u64 j = diag - 1 - mid;
        bool le = key_le<RANK>(p.a[mid], RANK ? p.ra[mid] : 0, p.b[j], RANK ? p.rb[j] : 0);
        if (le) lo = mid + 1; else hi = mid;
    }
    p.mp[t] = lo;
}

template <int OP, bool TAX, bool RANK, int VT>
__global__ __launch_bounds__(NT) void setop_tile_kernel(SetopArgs p) {
    constexpr int TILE = NT * VT;
    constexpr int SLOTS = TILE + 4;
    constexpr int LD = VT + 1;  // loads per thread to cover SLOTS
    __shared__ u64 s_keys[SLOTS];
    __shared__ u32 s_tax[TAX ? SLOTS : 1];
    __shared__ u32 s_rank[RANK ? SLOTS : 1];
    __shared__ u32 s_scan[NT / 64 + 1];
    __shared__ u64 s_misc[2];

    const int tid = (int)threadIdx.x;
    if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
    __syncthreads();
    const u64 tile = s_misc[0];

    const u64 N = p.na + p.nb;
    const u64 d0 = tile * (u64)TILE;
    const u64 d1 = (d0 + TILE < N) ? d0 + TILE : N;
    const u64 a0 = p.mp[tile], a1 = p.mp[tile + 1];
    const u64 b0 = d0 - a0, b1 = d1 - a1;
    const int na_t = (int)(a1 - a0), nb_t = (int)(b1 - b0);
    const int total = na_t + nb_t;
    const bool has_prev_a = a0 > 0, has_prev_b = b0 > 0, has_next_b = b1 < p.nb;
    // LDS slots: [0] prevA | A items [1, 1+na_t) | nextA | prevB | B items | nextB
    const int base_a = 1, end_a = 1 + na_t;
    const int base_b = na_t + 3, end_b = base_b + nb_t;

    {
        u64 rk[LD];
        u32 rt[LD];
        u32 rr[LD];
#pragma unroll
        for (int j = 0; j < LD; j++) {
            int i = tid + j * NT;
            u64 v = 0;
            u32 tv = 0, rv = 0;
            if (i < total + 4) {
                if (i < base_b - 1) {  // A region incl. both halos
                    long long g = (long long)a0 + (i - base_a);
                    if (g >= 0 && (u64)g < p.na) {
                        v = p.a[g];
                        if (TAX && p.ta) tv = p.ta[g];
                        if (RANK) rv = p.ra[g];
                    }
                } else {
                    long long g = (long long)b0 + (i - base_b);
                    if (g >= 0 && (u64)g < p.nb) {
                        v = p.b[g];
                        if (TAX && p.tb) tv = p.tb[g];
                        if (RANK) rv = p.rb[g];
                    }
                }
            }
            rk[j] = v;
            rt[j] = tv;
            rr[j] = rv;
        }
#pragma unroll
        for (int j = 0; j < LD; j++) {
            int i = tid + j * NT;
            if (i < SLOTS) {
                s_keys[i] = rk[j];
                if (TAX) s_tax[i] = rt[j];
                if (RANK) s_rank[i] = rr[j];
            }
        }
    }
    __syncthreads();

    // per-thread merge-path search inside the tile
    int diag = tid * VT;
    if (diag > total) diag = total;
    int lo = diag > nb_t ? diag - nb_t : 0;
    int hi = diag < na_t ? diag : na_t;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        int ia = base_a + mid, ib = base_b + diag - 1 - mid;
        bool le = key_le<RANK>(s_keys[ia], RANK ? s_rank[ia] : 0, s_keys[ib], RANK ? s_rank[ib] : 0);
        if (le) lo = mid + 1; else hi = mid;
    }
    int pa = base_a + lo, pb = base_b + diag - lo;

    u64 ak = s_keys[pa], bk = s_keys[pb];
    u64 ap = s_keys[pa - 1], bp = s_keys[pb - 1];
    u32 ar = 0, br = 0, apr = 0, bpr = 0;
    if (RANK) { ar = s_rank[pa]; br = s_rank[pb]; apr = s_rank[pa - 1]; bpr = s_rank[pb - 1]; }
    bool apv = (pa > base_a) || has_prev_a;
    bool bpv = (pb > base_b) || has_prev_b;

    u64 ok[VT];
    u32 ot[VT];
    u32 mask = 0, bad = 0;
    const bool mix = (p.flags & UKM_F_MIX_TAXID) != 0;
    const bool cmp = (p.flags & UKM_F_CMP_TAXID) != 0;

#pragma unroll
    for (int s = 0; s < VT; s++) {
        const bool a_ok = pa < end_a, b_ok = pb < end_b;
        const bool active = a_ok || b_ok;
        const bool take_a = a_ok && (!b_ok || key_le<RANK>(ak, ar, bk, br));
        bool emit = false;
        u64 ek = 0;
        u32 et = 0;
        if (active) {
            if (take_a) {
                if (apv) {  // strict order check of input A
                    bool lt = RANK ? (ap < ak || (ap == ak && apr < ar)) : (ap < ak);
                    if (!lt) bad |= (!RANK && ap == ak) ? FLAG_DUP : FLAG_UNSORTED;
                }
                const bool matched = (pb < end_b + (has_next_b ? 1 : 0)) && key_eq<RANK>(ak, ar, bk, br);
                ek = ak;
                u32 ta = 0, tb = 0;
                if (TAX) { ta = s_tax[pa]; tb = s_tax[pb]; }
                if (OP == UKM_OP_UNION) {
                    emit = true;
                    if (TAX) et = matched ? lca_dev(p.tax, ta, tb) : ta;
                } else if (OP == UKM_OP_INTER) {
                    emit = matched;
                    if (TAX && matched) {
                        if (mix) et = (ta == 0) ? tb : ((tb == 0) ? ta : lca_dev(p.tax, ta, tb));
                        else et = lca_dev(p.tax, ta, tb);
                    }
                } else {
                    emit = !matched;
                    if (TAX) {
                        et = ta;
                        if (matched && cmp && (ta == tb || lca_dev(p.tax, tb, ta) == ta)) emit = true;
                    }
                }
                ap = ak; apr = ar; apv = true;
                pa++;
            } else {
                if (bpv) {
                    bool lt = RANK ? (bp < bk || (bp == bk && bpr < br)) : (bp < bk);
                    if (!lt) bad |= (!RANK && bp == bk) ? FLAG_DUP : FLAG_UNSORTED;
                }
                if (OP == UKM_OP_UNION) {
                    const bool matched_prev = apv && key_eq<RANK>(ap, apr, bk, br);
                    emit = !matched_prev;
                    ek = bk;
                    if (TAX) et = s_tax[pb];
                }
                bp = bk; bpr = br; bpv = true;
                pb++;
            }
            // one LDS read refills whichever cursor moved
            const int idx = take_a ? pa : pb;
            const u64 nk = s_keys[idx];
            const u32 nr = RANK ? s_rank[idx] : 0;
            if (take_a) { ak = nk; ar = nr; } else { bk = nk; br = nr; }
        }
        ok[s] = ek;
        ot[s] = et;
        if (emit) mask |= (1u << s);
    }

    const u32 cnt = (u32)__popc(mask);
    u32 tile_total;
    const u32 excl = block_excl_scan_u32<NT>(cnt, s_scan, &tile_total);
    // (the scan's barriers also guarantee every thread finished reading the tile from LDS)

    {
        u32 w = excl;
#pragma unroll
        for (int s = 0; s < VT; s++) {
            if (mask & (1u << s)) {
                s_keys[w] = ok[s];
                if (TAX) s_tax[w] = ot[s];
                w++;
            }
        }
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
            p.out[pos] = s_keys[i];
            if (TAX) p.tout[pos] = s_tax[i];
        }
    }
    if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
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

template <int OP, bool TAX, bool RANK, int VT>
void launch_tile(const SetopArgs &p, hipStream_t st) {
    hipLaunchKernelGGL((setop_tile_kernel<OP, TAX, RANK, VT>), dim3((unsigned)p.ntiles), dim3(NT), 0, st, p);
}

template <bool TAX, bool RANK, int VT>
void launch_op(int op, const SetopArgs &p, hipStream_t st) {
    if (op == UKM_OP_UNION) launch_tile<UKM_OP_UNION, TAX, RANK, VT>(p, st);
    else if (op == UKM_OP_INTER) launch_tile<UKM_OP_INTER, TAX, RANK, VT>(p, st);
    else launch_tile<UKM_OP_DIFF, TAX, RANK, VT>(p, st);
}

constexpr int VT_PLAIN = 16;  // 4096-item tiles, 32 KiB of keys in LDS
constexpr int VT_TAX = 12;    // 3072-item tiles when taxids/ranks ride along

// One pass of the tiled set operation.  result_host[0] = total, [1] = flags.
int run_setop_pass(ukm_ctx *c, int op, const u64 *a, const u32 *ta, const u32 *ra, u64 na,
                   const u64 *b, const u32 *tb, const u32 *rb, u64 nb, bool tax, u32 flags,
                   u64 *out, u32 *tout, u64 out_cap, u64 result_host[2]) {
    const bool rank = ra != nullptr;
    const int vt = (tax || rank) ? VT_TAX : VT_PLAIN;
    const u64 tile_items = (u64)NT * vt;
    const u64 N = na + nb;
    SetopArgs p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.ta = ta; p.tb = tb; p.ra = ra; p.rb = rb;
    p.na = na; p.nb = nb;
    p.out = out; p.tout = tout; p.out_cap = out_cap;
    p.ntiles = (N + tile_items - 1) / tile_items;
    p.tax = ukm_taxdev(c);
    p.flags = flags;
    if (p.ntiles > 0xFFFFFFFFull) UKM_FAIL(UKM_ERR_INVALID, "setop: input too large");

    // control block: [result 2 x u64][ticket (u64 slot)][status ntiles][mp ntiles+1]
    u64 *ctl = nullptr;
    const size_t nzero = 3 + p.ntiles;
    UKM_TRY(ws_alloc_t(c, nzero + p.ntiles + 1, &ctl));
    p.result = ctl;
    p.ticket = (u32 *)(ctl + 2);
    p.status = ctl + 3;
    p.mp = ctl + 3 + p.ntiles;
    UKM_HIP(hipMemsetAsync(ctl, 0, nzero * sizeof(u64), c->stream));

    const unsigned pblocks = (unsigned)((p.ntiles + 1 + 255) / 256);
    if (rank)
        hipLaunchKernelGGL(setop_partition_kernel<true>, dim3(pblocks), dim3(256), 0, c->stream, p, (int)tile_items);
    else
        hipLaunchKernelGGL(setop_partition_kernel<false>, dim3(pblocks), dim3(256), 0, c->stream, p, (int)tile_items);

    (void)hipEventRecord(c->ev_k0, c->stream);
    if (rank) {
        if (tax) launch_op<true, true, VT_TAX>(op, p, c->stream);
        else launch_op<false, true, VT_TAX>(op, p, c->stream);
    } else {
        if (tax) launch_op<true, false, VT_TAX>(op, p, c->stream);
        else launch_op<false, false, VT_PLAIN>(op, p, c->stream);
    }
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    UKM_TRY(ukm_read_u64(c, p.result, result_host, 2));
    return UKM_OK;
}

}  // namespace

// internal entry: all pointers are device pointers
int ukm_dev_setop2(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u64 na, const u64 *b,
                   const u32 *tb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out) {
    if (op != UKM_OP_UNION && op != UKM_OP_INTER && op != UKM_OP_DIFF)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: unknown op %d", op);
    const bool tax = (ta != nullptr) || (tb != nullptr);
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: taxids given but out_taxids is NULL");
    const bool need_lca = tax && (op != UKM_OP_DIFF || (flags & UKM_F_CMP_TAXID));
    if (need_lca && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
    *n_out = 0;
    if (na + nb == 0) return UKM_OK;
    if (na + nb >= (1ull << 61)) UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: input too large");

    u64 res[2] = {0, 0};
    UKM_TRY(run_setop_pass(c, op, a, ta, nullptr, na, b, tb, nullptr, nb, tax, flags, out, tout,
                           out_cap, res));
    if (res[1] & FLAG_UNSORTED) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
    if (res[1] & FLAG_DUP) {
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
            UKM_TRY(ukm_dev_unique(c, a, ta, na, UKM_UNIQUE, ua, uta, na, &nua));
            UKM_TRY(ukm_dev_unique(c, b, tb, nb, UKM_UNIQUE, ub, utb, nb, &nub));
            UKM_TRY(run_setop_pass(c, op, ua, uta, nullptr, nua, ub, utb, nullptr, nub, tax, flags,
                                   out, tout, out_cap, res));
        } else {
            u32 *ra = nullptr, *rb = nullptr;
            UKM_TRY(ws_alloc_t(c, na + 1, &ra));
            UKM_TRY(ws_alloc_t(c, nb + 1, &rb));
            if (na) hipLaunchKernelGGL(rank_in_run_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, c->stream, a, na, ra);
            if (nb) hipLaunchKernelGGL(rank_in_run_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, b, nb, rb);
            if (op == UKM_OP_DIFF) {
                // the reference's survivor map collapses duplicate codes (diff.go:449-453);
                // the last record of a run wins
                u64 *tk = nullptr;
                u32 *tt = nullptr;
                UKM_TRY(ws_alloc_t(c, na + 1, &tk));
                if (tax) UKM_TRY(ws_alloc_t(c, na + 1, &tt));
                UKM_TRY(run_setop_pass(c, op, a, ta, ra, na, b, tb, rb, nb, tax, flags, tk, tt, na, res));
                if (!(res[1] & FLAG_UNSORTED)) {
                    u64 nu = 0;
                    UKM_TRY(ukm_dev_unique(c, tk, tax ? tt : nullptr, res[0], 4 /*UNIQUE_LAST*/, out, tout, out_cap, &nu));
                    res[0] = nu;
                }
            } else {
                UKM_TRY(run_setop_pass(c, op, a, ta, ra, na, b, tb, rb, nb, tax, flags, out, tout,
                                       out_cap, res));
            }
        }
        if (res[1] & FLAG_UNSORTED) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
    }
    *n_out = res[0];
    if (res[0] > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "ukm_setop2: output needs %llu records, capacity is %llu",
                 (unsigned long long)res[0], (unsigned long long)out_cap);
    return UKM_OK;
}

extern "C" int ukm_setop2(ukm_ctx *ctx, int op, const uint64_t *a_keys, const uint32_t *a_taxids,
                          uint64_t na, const uint64_t *b_keys, const uint32_t *b_taxids,
                          uint64_t nb, uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids,
                          uint64_t out_cap, uint64_t *n_out) {
    if (!ctx || !n_out || (!a_keys && na) || (!b_keys && nb) || (!out_keys && out_cap))
        UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: NULL argument");
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
        int r = ukm_dev_setop2(ctx, op, a, ta, na, b, tb, nb, flags, out, tout, out_cap, n_out);
        u64 n = (r == UKM_OK) ? *n_out : 0;
        ukm_out_resize(ctx, out_keys, n * sizeof(u64));
        if (out_taxids) ukm_out_resize(ctx, out_taxids, n * sizeof(u32));
        return r;
    }();
    return ukm_finish(&s, rc);
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
