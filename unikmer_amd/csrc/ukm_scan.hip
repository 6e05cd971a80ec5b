// ukm_scan.hip — scans over a SORTED (code [, taxid]) stream: the replacement for the linear
// dedup / LCA-fold / "repeated" loops of sort.go:484-572 and dumpCodes[Taxids]2File
// (util-sort.go:35-190).  Plus two small primitives used across the library: sortedness check
// and a single-pass exclusive scan of uint64.
//
// Unique kernel: one 512-thread workgroup per tile of 8192 records, striped across the threads;
// keys are loaded coalesced into LDS with a one-element halo on both sides; run heads are found
// by adjacent comparison (conflict-free neighbour reads); each head emits 0, 1 or 2 records
// according to the mode; the per-code taxid is the LCA fold over the run (associative +
// commutative on a tree, so fold order is immaterial); positions come from one wave64 ballot
// per round plus a 128-entry prefix per tile, compaction is in place, and the global offset
// from the same single-pass decoupled look-back as the set operations: the stream is read
// once and written once: 8n (+4n) bytes in, 8u (+4u) bytes out.
#include <algorithm>

#include "ukm_device.h"

#define UKM_UNIQUE_LAST 5  // internal: one record per run, the LAST record of the run
#define UKM_COMMON 6       // internal: run heads whose run length >= threshold (common.go:331-335)

namespace {

constexpr int NT = 256;   // exclusive-scan kernel
#ifndef UNIQ_NT
#define UNIQ_NT 512
#endif
#ifndef UNIQ_VT
#define UNIQ_VT 16
#endif
constexpr int UNT = UNIQ_NT;  // unique kernel: threads per workgroup
constexpr int UVT = UNIQ_VT;  // records per thread (half of it for the chunk protocol)
#ifndef UNIQ_VT_TAX
#define UNIQ_VT_TAX 12
#endif
constexpr int UVT_TAX = UNIQ_VT_TAX;  // with taxids: 12 B per staged record, 80 KB of LDS -> two workgroups per CU instead of one

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

// Striped layout: thread t handles records t, t+NT, t+2NT, ... of the tile, so that
//   * global loads are coalesced and the prev/next neighbours of a record sit in the adjacent
//     LDS slots read by the adjacent lanes (bank-conflict free; the first version read 16-slot
//     strided blocks per thread: 8-16-way conflicts),
//   * an emitted record's position is (records emitted in earlier (round, wave) groups) +
//     (emitting lower lanes in its own wave) = one wave64 ballot per round plus one 128-entry
//     prefix per tile — no per-thread output arrays, compaction in place over the input tile.
// CHUNK = the REPEATED_CHUNK protocol (up to two output records per input record).
// TICKET: tile ids from an atomic counter (always live) instead of blockIdx (no single-address atomic in front
// of every tile, but look-back liveness then relies on in-order dispatch: watchdog -> flag 4 -> the host re-runs
// the ticketed instantiation, see ukm_setops.hip).
#ifndef UNIQ_OWN_N
#define UNIQ_OWN_N 8
#endif
constexpr int UNIQ_OWN = UNIQ_OWN_N;  // records of a run its head's lane folds itself before the wave takes over
constexpr int UNIQ_FEW = 4;  // ... one head at a time, when at most this many lanes of the wave ask for it

template <bool TAX, bool CHUNK, int VTU, bool TICKET>
__global__ __launch_bounds__(UNT) void unique_tile_kernel(UniqArgs p) {
    constexpr int TILE_U = UNT * VTU;
    constexpr int NWU = UNT / 64;
    constexpr int OUT_SLOTS = CHUNK ? 2 * TILE_U : TILE_U;
    __shared__ __attribute__((aligned(16))) u64 s_keys[(OUT_SLOTS > TILE_U + 2 ? OUT_SLOTS : TILE_U + 2)];
    __shared__ u32 s_tax[TAX ? OUT_SLOTS : 1];
    __shared__ u32 s_cnt[VTU * NWU + 1];
    __shared__ u64 s_misc[2];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    u64 tile = blockIdx.x;
    if (TICKET) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    }
    const u64 i0 = tile * (u64)TILE_U;
    const int cnt_t = (int)((p.n - i0 < (u64)TILE_U) ? (p.n - i0) : (u64)TILE_U);

    // slot j holds k[i0 + j - 1]: halo | tile | halo
    u64 key[VTU];
#pragma unroll
    for (int s = 0; s < VTU; s++) {
        const int j = tid + s * UNT;
        key[s] = (j < cnt_t) ? p.k[i0 + j] : 0;
    }
    if (tid == 0) s_keys[0] = (i0 > 0) ? p.k[i0 - 1] : 0;
    if (tid == 64) s_keys[cnt_t + 1] = (i0 + cnt_t < p.n) ? p.k[i0 + cnt_t] : 0;
#pragma unroll
    for (int s = 0; s < VTU; s++) {
        const int j = tid + s * UNT;
        if (j < cnt_t) s_keys[j + 1] = key[s];
    }
    __syncthreads();

    const int mode = p.mode;
    u32 bad = 0;
    u32 emit1 = 0, emit2 = 0;  // bit s: record of round s is emitted once / a second time
    u32 tx[TAX ? VTU : 1];
#pragma unroll
    for (int s = 0; s < VTU; s++) {
        const int j = tid + s * UNT;
        const bool in = j < cnt_t;
        const u64 gi = i0 + (u64)j;
        const u64 k = key[s];
        const u64 prev = s_keys[in ? j : 0], next = s_keys[in ? j + 2 : 0];
        const bool has_prev = gi > 0, has_next = gi + 1 < p.n;
        if (in && has_prev && prev > k) bad |= 2;
        const bool head = in && (!has_prev || prev != k);
        const bool tail = in && (!has_next || next != k);
        const bool repeated = !tail;
        bool e1 = false, e2 = false;
        if (mode == UKM_UNIQUE) e1 = head;
        else if (mode == UKM_REPEATED) e1 = head && repeated;
        else if (mode == UKM_SINGLETON) e1 = head && !repeated;
        else if (mode == UKM_REPEATED_CHUNK) { e1 = head; e2 = head && repeated; }
        else if (mode == UKM_UNIQUE_LAST) e1 = tail;
        u32 t = 0;
        if (mode == UKM_COMMON && head) {
            // run length against the threshold (common.go:331-335): in a sorted sequence the run has `threshold` records
            // iff the record threshold - 1 places on carries the same code — one probe, whatever the run length
            const u64 need = p.threshold > 1 ? (u64)p.threshold - 1 : 0;
            e1 = need == 0 || (gi + need < p.n && p.k[gi + need] == k);
        }
        if (TAX) {
            // LCA over the run of an emitted record (common.go:265, sort.go:491).  The lane folds up to UNIQ_OWN records
            // itself; what a longer run has behind them (a merge of n files: runs of up to n records) is folded by the
            // whole wave, 64 records per step, through the pre-order numbers (ukm_internal.h: TaxDev::euler): the
            // sequential fold of lca_dev over a set is its first member when all are equal, 0 when they differ and one is
            // unknown, else the LCA of the members with the smallest and the largest number.
            bool more = false;
            u64 qn = 0;
            if (mode == UKM_UNIQUE_LAST) {
                if (e1) t = p.t[gi];
            } else if (e1) {
                t = p.t[gi];
                if (mode == UKM_COMMON || repeated) {
                    if (p.tax.euler) {
                        // the codes behind the head come from the staged tile (global memory behind its end), their
                        // taxids are fetched together: one round trip + one per LCA instead of three per record
                        int own = 0;
                        bool open = true;
#pragma unroll
                        for (int i = 0; i <= UNIQ_OWN; i++) {
                            const u64 q = gi + 1 + (u64)i;
                            u64 kq = ~k;
                            if (j + 1 + i < cnt_t) kq = s_keys[j + 2 + i];
                            else if (q < p.n) kq = p.k[q];
                            open = open && kq == k;
                            if (i < UNIQ_OWN) own += open ? 1 : 0;
                            else more = open;
                        }
                        u32 tq[UNIQ_OWN];
#pragma unroll
                        for (int i = 0; i < UNIQ_OWN; i++) tq[i] = p.t[gi + (u64)(i < own ? 1 + i : 0)];
#pragma unroll
                        for (int i = 0; i < UNIQ_OWN; i++)
                            if (i < own) t = lca_dev(p.tax, tq[i], t);
                        qn = gi + 1 + (u64)own;
                    } else {
                        u64 q = gi + 1;
                        for (; q < p.n && p.k[q] == k; q++) t = lca_dev(p.tax, p.t[q], t);
                    }
                }
            }
            if (__popcll(__ballot(more)) > UNIQ_FEW) {
                // many heads of this wave want more: runs of a few dozen records one after the other (chunk files of the
                // same genomes).  The wave-wide fold takes them one at a time; their lanes walking on side by side is
                // cheaper up to a wave's width of records.
                if (more) {
                    int own = UNIQ_OWN;
                    for (; own < 64 && qn < p.n && p.k[qn] == k; qn++, own++) t = lca_dev(p.tax, p.t[qn], t);
                    more = own == 64 && qn < p.n && p.k[qn] == k;
                }
            }
            for (u64 m = __ballot(more); m != 0ull; m &= m - 1ull) {
                const int lead = __ffsll((long long)m) - 1;
                const u64 k0 = ((u64)(u32)__shfl((int)(k >> 32), lead, 64) << 32) | (u32)__shfl((int)(u32)k, lead, 64);
                u64 q0 = ((u64)(u32)__shfl((int)(qn >> 32), lead, 64) << 32) | (u32)__shfl((int)(u32)qn, lead, 64);
                const u32 t0 = (u32)__shfl((int)t, lead, 64);
                const u32 e0 = t0 < p.tax.size ? p.tax.euler[t0] : 0u;
                u32 mn = 0xFFFFFFFFu, mx = 0u, fl = 0u;  // fl: 1 = a taxid differs from t0, 2 = one of those is unknown
                for (;;) {
                    const u64 pos = q0 + (u64)lane;
                    const u64 pc = pos < p.n ? pos : p.n - 1;
                    const u64 kk = p.k[pc];
                    const u32 tt = p.t[pc];
                    const bool match = pos < p.n && kk == k0;
                    if (match && tt != t0) {
                        const u32 e = tt < p.tax.size ? p.tax.euler[tt] : 0u;
                        fl |= e ? 1u : 3u;
                        if (e) {
                            mn = e < mn ? e : mn;
                            mx = e > mx ? e : mx;
                        }
                    }
                    if (__ballot(match) != ~0ull) break;
                    q0 += 64;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const u32 omn = (u32)__shfl_xor((int)mn, d, 64), omx = (u32)__shfl_xor((int)mx, d, 64);
                    mn = omn < mn ? omn : mn;
                    mx = omx > mx ? omx : mx;
                    fl |= (u32)__shfl_xor((int)fl, d, 64);
                }
                if (lane == lead && (fl & 1u)) {
                    if ((fl & 2u) || !e0) t = 0;
                    else t = lca_dev(p.tax, p.tax.node_at[e0 < mn ? e0 : mn], p.tax.node_at[e0 > mx ? e0 : mx]);
                }
            }
        }
        if (TAX) tx[s] = t;
        emit1 |= e1 ? (1u << s) : 0u;
        emit2 |= e2 ? (1u << s) : 0u;
    }
    // per (round, wave) output counts; records are ordered round-major, then wave, then lane
    u32 pos_in_group[VTU];
#pragma unroll
    for (int s = 0; s < VTU; s++) {
        const bool e1 = (emit1 >> s) & 1u, e2 = (emit2 >> s) & 1u;
        const u64 m1 = __ballot(e1);
        const u64 lt = (1ull << lane) - 1;
        u32 before = (u32)__popcll(m1 & lt), total = (u32)__popcll(m1);
        if (CHUNK) {
            const u64 m2 = __ballot(e2);
            before += (u32)__popcll(m2 & lt);
            total += (u32)__popcll(m2);
        }
        pos_in_group[s] = before;
        if (lane == 0) s_cnt[s * NWU + wave] = total;
    }
    __syncthreads();  // also: every thread is done reading the input tile from LDS
    if (tid < 64) {   // exclusive prefix over the VTU*NWU groups (<= 128 entries: two per lane)
        constexpr int NG = VTU * NWU;
        const int e0 = 2 * lane, e1i = 2 * lane + 1;
        const u32 c0 = e0 < NG ? s_cnt[e0] : 0, c1 = e1i < NG ? s_cnt[e1i] : 0;
        const u32 incl = wave_incl_scan_u32(c0 + c1);
        const u32 ex = incl - (c0 + c1);
        if (e0 < NG) s_cnt[e0] = ex;
        if (e1i < NG) s_cnt[e1i] = ex + c0;
        if (lane == 63) s_cnt[NG] = incl;
    }
    __syncthreads();
    const u32 tile_total = s_cnt[VTU * NWU];
    if (tid == 0) lb_publish(p.status, tile, (u64)tile_total);
#pragma unroll
    for (int s = 0; s < VTU; s++) {
        const bool e1 = (emit1 >> s) & 1u, e2 = (emit2 >> s) & 1u;
        if (e1) {
            const u32 w = s_cnt[s * NWU + wave] + pos_in_group[s];
            s_keys[w] = key[s];
            if (TAX) s_tax[w] = tx[s];
            if (CHUNK && e2) {
                s_keys[w + 1] = key[s];
                if (TAX) s_tax[w + 1] = tx[s];
            }
        }
    }
    if (tid < 64) {
        bool timed_out = false;
        const u64 base = lb_resolve(p.status, tile, (u64)tile_total, lane, TICKET ? nullptr : &timed_out);
        if (tid == 0) s_misc[1] = base;
        if (timed_out) bad |= 4u;
    }
    if (bad) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)bad);
    __syncthreads();
    const u64 base = s_misc[1];
    if (base + tile_total <= p.out_cap) {
        u64 *o = p.out + base;
        const int sh = (int)(((uintptr_t)o >> 3) & 1);
        const int npairs = ((int)tile_total + sh + 1) >> 1;
        for (int m = tid; m < npairs; m += UNT) {
            const int j0 = 2 * m - sh, j1 = j0 + 1;
            const bool v0 = j0 >= 0, v1 = j1 < (int)tile_total;
            const u64 k0 = s_keys[v0 ? j0 : 0], k1 = s_keys[v1 ? j1 : 0];
            if (v0 && v1) *reinterpret_cast<ulonglong2 *>(o + j0) = make_ulonglong2(k0, k1);
            else if (v0) o[j0] = k0;
            else if (v1) o[j1] = k1;
        }
        if (TAX) {
            u32 *to = p.tout + base;
            for (u32 i = (u32)tid; i < tile_total; i += UNT) to[i] = s_tax[i];
        }
    } else {
        for (u32 i = (u32)tid; i < tile_total; i += UNT) {
            const u64 pos = base + i;
            if (pos < p.out_cap) {
                p.out[pos] = s_keys[i];
                if (TAX) p.tout[pos] = s_tax[i];
            }
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

// dst[i] = value (or *value_dev when that is given: a taxid an earlier kernel of the stream worked out), 16 bytes per store
// on the aligned middle
__global__ void fill_u32_kernel(u32 *dst, u64 n, u32 value, const u32 *value_dev) {
    const u32 v = value_dev ? *value_dev : value;
    const u64 mis = (u64)((4 - (((uintptr_t)dst >> 2) & 3)) & 3);
    const u64 head = n < mis ? n : mis;
    const u64 nq = (n - head) / 4;
    uint4 *q = reinterpret_cast<uint4 *>(dst + head);
    const uint4 vv = make_uint4(v, v, v, v);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) q[i] = vv;
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) dst[threadIdx.x] = v;
        const u64 tail = head + 4 * nq;
        if (tail + threadIdx.x < n && threadIdx.x < 4) dst[tail + threadIdx.x] = v;
    }
}

}  // namespace

int ukm_dev_fill_u32(ukm_ctx *c, u32 *dst, u64 n, u32 value) { return ukm_dev_fill_u32_from(c, dst, n, value, nullptr); }

int ukm_dev_fill_u32_from(ukm_ctx *c, u32 *dst, u64 n, u32 value, const u32 *value_dev) {
    if (n == 0) return UKM_OK;
    const unsigned blocks = (unsigned)std::min<u64>((n / 4 + 255) / 256 + 1, (u64)c->num_cu * 16);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, c->stream, dst, n, value, value_dev);
    UKM_HIP(hipGetLastError());
    return UKM_OK;
}

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
    const bool chunk = mode == UKM_REPEATED_CHUNK;
    const int vt = tax ? UVT_TAX : UVT;
    const u64 tile_items = (u64)UNT * (chunk ? vt / 2 : vt);
    p.ntiles = (n + tile_items - 1) / tile_items;
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
    const dim3 grid((unsigned)p.ntiles), block(UNT);
    // a first attempt that times out may already have written part of the output: when the output aliases
    // the input the retry would read damaged data, so in-place calls take the ticketed kernel straight away
    const bool in_place = (out <= keys && keys < out + out_cap) || (keys <= out && out < keys + n);
    u64 res[2] = {0, 0};
    for (int attempt = (c->setop_force_ticket || in_place) ? 1 : 0; attempt < 2; attempt++) {
        if (attempt == 1) UKM_HIP(hipMemsetAsync(ctl, 0, nctl * sizeof(u64), c->stream));
        if (attempt == 0) {
            if (chunk) {
                if (tax) hipLaunchKernelGGL((unique_tile_kernel<true, true, UVT_TAX / 2, false>), grid, block, 0, c->stream, p);
                else hipLaunchKernelGGL((unique_tile_kernel<false, true, UVT / 2, false>), grid, block, 0, c->stream, p);
            } else {
                if (tax) hipLaunchKernelGGL((unique_tile_kernel<true, false, UVT_TAX, false>), grid, block, 0, c->stream, p);
                else hipLaunchKernelGGL((unique_tile_kernel<false, false, UVT, false>), grid, block, 0, c->stream, p);
            }
        } else {
            if (chunk) {
                if (tax) hipLaunchKernelGGL((unique_tile_kernel<true, true, UVT_TAX / 2, true>), grid, block, 0, c->stream, p);
                else hipLaunchKernelGGL((unique_tile_kernel<false, true, UVT / 2, true>), grid, block, 0, c->stream, p);
            } else {
                if (tax) hipLaunchKernelGGL((unique_tile_kernel<true, false, UVT_TAX, true>), grid, block, 0, c->stream, p);
                else hipLaunchKernelGGL((unique_tile_kernel<false, false, UVT, true>), grid, block, 0, c->stream, p);
            }
        }
        UKM_HIP(hipGetLastError());
        UKM_TRY(ukm_read_u64(c, p.result, res, 2));
        if (!(res[1] & 4)) break;
        if (attempt == 1) UKM_FAIL(UKM_ERR_HIP, "ukm_unique: look-back watchdog fired in the ticketed kernel");
        ukm_switch_to_tickets(c, "unique kernel");
    }
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
