// ukm_pfold.hip — `inter` / `diff` over MANY sorted sets by LDS hash probes (round 3, after ukm_punion.hip).
// Replaces the per-file loops inter.go:205-286 / diff.go:379-454 like ukm_fold.hip does, for the rules that do not depend
// on the ORDER of the files:
//     inter        a code of file 0 survives when every later file has it; TaxId = LCA over all files (inter.go:229-239
//                  without --mix-taxid: LCA is associative and commutative, 0 / unknown absorbing — ukm_device.h lca_dev)
//     diff         a code of file 0 survives when no later file has it; it keeps its own TaxId (diff.go:404-409)
//     diff -t      ... when no later file has it with a taxid that differs from its own and does not lie below it
// (`inter --mix-taxid` goes to the range fold of ukm_fold.hip, whose survivors see the files in order; so does, by
//  default, `inter` with taxids: see ukm_dev_probe_fold.)
//
// ukm_fold.hip keeps the survivors in registers and looks each of them up in every file's slice (a lock-step binary
// search per survivor per file: bound by dependent-instruction latency, 1.6-1.9 TB/s).  Here the roles are swapped, as in
// ukm_punion.hip: file 0 is cut into ranges of L <= 1536 records; one workgroup per range puts its records into a
// bucketised LDS table (1024 buckets x 4 slots, slot -> record index beside it) and its waves STREAM the slices of the
// later files, one slice per wave at a time: per record one hash and one 32-byte bucket read; a hit bumps the record's
// counter (inter: alive = S - 1 hits; diff: any hit kills) and, for inter with taxids, folds the TaxId in with a CAS
// loop (skipped for records that are already known to be dead: fewer hits than files finished).  Every file is
// checked for strictly increasing order on the way; a duplicate code anywhere (or an all-ones code in file 0, which
// is the table's empty marker) sends the call to the exact routes behind it.
// Algorithmic bytes: 8 per record of the later files (+ 4 for inter with taxids), read once.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "ukm_device.h"
#include "ukm_pfold.h"

namespace {

constexpr int PF_NT = 512;
constexpr int PF_BUCKET_BITS = 10;
constexpr int PF_BUCKETS = 1 << PF_BUCKET_BITS;
constexpr int PF_SLOTS = 4 * PF_BUCKETS;
constexpr int PF_PER = 3;                // records of file 0 per thread (range length <= 1536; three workgroups per CU)

#ifndef PF_NT_EULER_N
#define PF_NT_EULER_N 1024
#endif
constexpr int PF_NT_EULER = PF_NT_EULER_N;  // inter with taxids: 16 waves per workgroup, two workgroups per CU = 8 waves per SIMD
constexpr int PF_MAXL = PF_NT * PF_PER;
constexpr u32 PF_CNT_MASK = 0x3FFFFFFFu, PF_NEQ = 0x40000000u, PF_BAD = 0x80000000u;  // inter + taxids: flags in the counter word
constexpr int PF_MINL = 256;
constexpr u64 PF_EMPTY = ~0ull;
enum { PF_FLAG_DUP = 1, PF_FLAG_UNSORTED = 2 };

struct PfArgs {
    const u64 *tab;  // [keys S][taxids S][lens S] (device copies of the caller's tables)
    u32 S, R, L;
    u64 *cuts;       // [R + 1][S - 1]: lower bound of file0[r * L] in file j (j = 1 .. S - 1)
    u64 *tmp_k;      // [R][L]
    u32 *tmp_t;
    u64 *cnt;        // [R]
    u64 *ctl;        // [0] total (written by the scan), [1] flags
    TaxDev T;
};

__device__ __forceinline__ u32 pf_hash(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return ((lo ^ __builtin_rotateleft32(hi, 15) ^ (hi >> 3)) * 0x9E3779B1u) >> (32 - PF_BUCKET_BITS);
}

__global__ void pf_cuts_kernel(PfArgs a) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 per = (u64)a.R + 1;
    const u32 S1 = a.S - 1;
    if (gid >= per * S1) return;
    const u32 j = (u32)(gid / per) + 1, r = (u32)(gid % per);
    const u64 len = a.tab[2 * (u64)a.S + j];
    u64 res;
    if (r == 0) {
        res = 0;
    } else if (r == a.R) {
        res = len;
    } else {
        const auto f0 = as_global((const u64 *)(uintptr_t)a.tab[0]);
        const auto f = as_global((const u64 *)(uintptr_t)a.tab[j]);
        const u64 v = f0[(u64)r * a.L];
        u64 lo = 0, hi = len;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (f[mid] < v) lo = mid + 1; else hi = mid;
        }
        res = lo;
    }
    a.cuts[(u64)r * S1 + (j - 1)] = res;
}

typedef u64 pf_u64x2 __attribute__((ext_vector_type(2)));
typedef pf_u64x2 __attribute__((aligned(8))) pf_pair;
typedef u32 pf_u32x2 __attribute__((ext_vector_type(2)));
typedef pf_u32x2 __attribute__((aligned(4))) pf_tpair;

// (three 512-thread workgroups per CU = 6 waves per SIMD: 85 registers; inter with taxids: 72 KB of LDS, two per CU)
template <int OP, bool TAX, bool CMP = false>
__global__ __launch_bounds__((OP == UKM_OP_INTER && TAX) ? PF_NT_EULER : PF_NT)
__attribute__((amdgpu_waves_per_eu((OP == UKM_OP_INTER && TAX) ? PF_NT_EULER / 128 : 6, (OP == UKM_OP_INTER && TAX) ? PF_NT_EULER / 128 : 6)))
void pf_probe_kernel(PfArgs a) {
    __shared__ __attribute__((aligned(32))) u64 s_tab[PF_SLOTS];
    __shared__ unsigned short s_idx[PF_SLOTS];
    constexpr bool EULER = OP == UKM_OP_INTER && TAX;  // inter with taxids: LCA of all files' taxids from pre-order numbers
    constexpr int NT = EULER ? PF_NT_EULER : PF_NT;
    constexpr int PER = EULER ? 2048 / PF_NT_EULER : PF_PER, MAXL = NT * PER;
    // inter with taxids keeps a record's four words side by side (round 6: x = counter + flags, y = its own taxid, z / w = the
    // smallest / largest number folded so far): a hit reads them with ONE 16-byte LDS read instead of four or five reads
    __shared__ u32 s_cnt[EULER ? 1 : MAXL];
    __shared__ u32 s_tax[(TAX && !EULER) ? MAXL : 1];
    __shared__ uint4 s_st4[EULER ? MAXL : 1];
    auto CNT = [&](u32 i) -> u32 & { if constexpr (EULER) return s_st4[i].x; else return s_cnt[i]; };
    auto TAXR = [&](u32 i) -> u32 & { if constexpr (EULER) return s_st4[i].y; else return s_tax[i]; };
    __shared__ u32 s_scan[NT / 64 + 1];
    __shared__ u32 s_next, s_done, s_dead;  // s_dead: no record of the range can survive any more
    constexpr bool FOLD_TAX = TAX && (OP == UKM_OP_INTER || CMP);  // the later files' taxids are read
    const int tid = (int)threadIdx.x, lane = lane_id();
    const u32 r = blockIdx.x, S = a.S, S1 = S - 1, L = a.L;
    const auto f0 = as_global((const u64 *)(uintptr_t)sload_u64(&a.tab[0]));
    const auto t0 = as_global((const u32 *)(uintptr_t)sload_u64(&a.tab[S]));
    const u64 len0 = sload_u64(&a.tab[2 * (u64)S]);
    const u64 e0 = (u64)r * L;
    const u32 ne = (u32)((len0 - e0 < (u64)L) ? (len0 - e0) : (u64)L);
    for (int i = tid; i < PF_SLOTS; i += NT) s_tab[i] = PF_EMPTY;
    if (tid == 0) { s_next = 0; s_done = 0; s_dead = 0; }
    u32 flags = 0;
    u64 ent[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)tid + (u32)k * NT;
        ent[k] = PF_EMPTY;
        if (i < ne) {
            const u64 e = f0[e0 + i];
            ent[k] = e;
            CNT(i) = 0;
            if (TAX) {
                const u32 tf = t0 ? t0[e0 + i] : 0u;
                TAXR(i) = tf;
                if constexpr (EULER) {
                    u32 ef = tf < a.T.size ? a.T.euler[tf] : 0u;
                    // (with one-byte clade codes the words hold clade << 24 | number: see the step)
                    if (ef && a.T.clade8) ef |= (u32)a.T.clade8[tf] << 24;
                    s_st4[i].z = ef ? ef : 0xFFFFFFFFu;
                    s_st4[i].w = ef;
                    if (!ef) s_st4[i].x = PF_BAD;
                }
            }
            if (e0 + i + 1 < len0) {  // file 0 strictly increasing (also across the range's end)
                const u64 nx = f0[e0 + i + 1];
                if (e == nx) flags |= PF_FLAG_DUP;
                if (e > nx) flags |= PF_FLAG_UNSORTED;
            }
            if (e == PF_EMPTY) flags |= PF_FLAG_DUP;  // (the empty marker: the exact route takes the call)
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)tid + (u32)k * NT;
        const u64 e = ent[k];
        if (i >= ne || e == PF_EMPTY) continue;
        // first free slot of the first bucket of its probe sequence that is not full (slots fill in order, nothing is
        // ever removed: "slot 3 taken" = "bucket full")
        u32 h = pf_hash(e);
        for (bool placed = false; !placed; h = (h + 1) & (PF_BUCKETS - 1)) {
#pragma unroll
            for (int q = 0; q < 4 && !placed; q++) {
                const u64 old = atomicCAS((unsigned long long *)&s_tab[4 * h + q], (unsigned long long)PF_EMPTY, (unsigned long long)e);
                if (old == PF_EMPTY) { s_idx[4 * h + q] = (unsigned short)i; placed = true; }
                else if (old == e) { flags |= PF_FLAG_DUP; placed = true; }
            }
        }
    }
    __syncthreads();

    // record index of x in this range's table, or -1
    auto find = [&](u64 x) -> int {
        u32 h = pf_hash(x);
        for (;;) {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(&s_tab[4 * h]);
            const ulonglong2 p = b[0], q = b[1];
            const bool m0 = p.x == x, m1 = p.y == x, m2 = q.x == x, m3 = q.y == x;
            if (m0 | m1 | m2 | m3) {
                if (x == PF_EMPTY) return -1;
                const int k = m0 ? 0 : (m1 ? 1 : (m2 ? 2 : 3));
                return (int)s_idx[4 * h + k];
            }
            if (q.y == PF_EMPTY) return -1;
            h = (h + 1) & (PF_BUCKETS - 1);
        }
    };
    auto hit = [&](int idx, u32 tb) {
        if (OP == UKM_OP_DIFF) {
            if (CMP) {
                // diff -t (diff.go:404-409): the hit removes the code unless the file's taxid equals the survivor's own or
                // lies below it.  The survivor's taxid never changes, so the files may come in any order; a code that is
                // already gone needs no LCA (with taxids that rarely nest that is nearly every hit after the first).
                if (CNT(idx) == 0) {
                    const u32 ta = TAXR(idx);
                    const bool keep = ta == tb || lca_dev(a.T, tb, ta) == ta;
                    if (!keep) CNT(idx) = 1;
                }
            } else {
                CNT(idx) = 1;  // (every writer stores the same value)
            }
        } else {
            atomicAdd(&CNT(idx), 1u);  // (inter with taxids does not come here: see the step)
        }
    };
    // N records of one later file (from registers; v[q]: the record is one of the slice's): hits are counted / folded
    auto process = [&](auto NN, const u64 *x, const u32 *tb, const bool *v) {
        constexpr int N = decltype(NN)::value;
        if constexpr (OP == UKM_OP_INTER && TAX) {
            // inter with taxids: count the hits now, fold the taxids below with the table reads of all of the
            // step's LCAs in flight together
            int li[N];
#pragma unroll
            for (int q = 0; q < N; q++) li[q] = v[q] ? find(x[q]) : -1;
            const u32 *lt = tb;
            // The LCA of a SET of taxids is the LCA of its members with the smallest and the largest pre-order number
            // (TaxDev::euler): a hit only has to fold its taxid's number into the record's minimum and maximum — two
            // commutative LDS atomics, files in any order — and ONE table LCA per survivor follows at the end.  Taxid 0 /
            // unknown ids (number 0) make the result 0 (lca_dev: absorbing) unless every taxid of the record is the
            // same (lca_dev: a == b -> a), which the NEQ flag keeps track of.
            u32 en[N];
            const bool byc = a.T.clade8 != nullptr;  // (uniform over the launch)
            // hits whose taxid has to be folded, as a bit mask.  (Marking the others by li[q] = -1, as rounds 3-4 did, let a hit
            // that carries the record's own taxid through to the fold below once that fold had two branches: such hits came
            // with the table's entry for taxid 0 and voided the record -- found by test_range_fold_equals_chained_fold_large)
            u32 needm = 0;
            // files finished BEFORE these hits are counted (acquire: the counts are not moved in front of the read): a record
            // that is in every file has one hit from each of them in its counter by now, so a counter below that number
            // belongs to a record some finished file did not have -- dead.  Read once per step: an older (smaller) number only
            // lets a few dead records through to the fold, whose words nobody looks at.
            const u32 finished = __hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int q = 0; q < N; q++) {
                bool need = false;
                if (li[q] >= 0) {
                    const uint4 st = s_st4[li[q]];
                    // (a record that is already behind -- most hits in a collection's non-core codes are on such records --
                    //  needs no count any more; one that is not gets its count WITHOUT a return value: it can only grow, so
                    //  it is above `finished` afterwards whatever the other waves add)
                    if ((st.x & PF_CNT_MASK) >= finished) {
                        __hip_atomic_fetch_add(&s_st4[li[q]].x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        need = lt[q] != st.y;
                    }
                }
                needm |= need ? (1u << q) : 0u;
                const u32 tq_ = (need && lt[q] < a.T.size) ? lt[q] : 0u;
                // (all of the step's table reads in flight; euler[0] = 0, clade8[0] = 0)
                en[q] = byc ? (u32)a.T.clade8[tq_] : a.T.euler[tq_];
            }
#pragma unroll
            for (int q = 0; q < N; q++) {
                if (!((needm >> q) & 1u)) continue;
                u32 *w = &s_st4[li[q]].x;
                if (!byc) {
                    atomicOr(w, en[q] ? PF_NEQ : (PF_NEQ | PF_BAD));
                    if (en[q]) {
                        atomicMin(w + 2, en[q]);
                        atomicMax(w + 3, en[q]);
                    }
                    continue;
                }
                // Round 5: a taxid's CLADE code (one byte of a table that stays in L2; codes are handed out in pre-order, so
                // they order taxids as the numbers do) in the top byte of the two words.  A record whose taxids span two
                // clades has the LCA of those two clade nodes whatever the exact numbers are: the 4-byte number of a
                // taxid (a random read of a table of 4 B x ids) is only fetched while the record's interval lies inside
                // ONE clade and the taxid is of that clade -- for taxids that are not related, next to never -- and a taxid
                // inside an interval of several clades costs one LDS read and no atomic at all.  (The interval only ever
                // widens: a record that is still inside one clade at the end has had every one of its taxids folded exactly.)
                const u32 c = en[q];
                if (c == 0) { atomicOr(w, PF_NEQ | PF_BAD); continue; }
                const uint4 st = s_st4[li[q]];
                if (!(st.x & PF_NEQ)) atomicOr(w, PF_NEQ);
                const u32 cmn = st.z >> 24, cmx = st.w >> 24;
                if (cmn == cmx && c == cmn) {
                    const u32 e = (c << 24) | a.T.euler[lt[q]];
                    atomicMin(w + 2, e);
                    atomicMax(w + 3, e);
                } else {
                    if (c < cmn) atomicMin(w + 2, (c << 24) | 0xFFFFFFu);
                    if (c > cmx) atomicMax(w + 3, c << 24);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < N; q++)
                if (v[q]) {
                    const int i0 = find(x[q]);
                    if (i0 >= 0) hit(i0, tb[q]);
                }
        }
    };
    // The streaming skeleton of pu2_probe_kernel (ukm_punion.hip, round 6).  A STEP = up to 128 records [lo, hi) of the 128
    // at `ptr` (two per lane; lanes whose pair lies beyond hi - 2 re-read the last pair that fits).  A slice that does not
    // begin its file starts one record early (lo = 1): the pair (f[beg - 1], f[beg]) is then checked inside lane 0 like every
    // other pair, and a last step of one record is moved back by one record the same way -- every load is a 16-byte pair
    // inside the file, and no third load per lane fetches the record behind a pair: its predecessor is the neighbouring
    // lane's second record (DPP wave_shr:1), lane 0 takes the previous step's last record from a scalar.  A slice = one
    // general first step, batches of full steps (no validity masks), general steps for what is left.
    const u32 l2 = 2u * (u32)lane;
    u64 run_carry = 0, ptr = 0, tptr = 0;
    bool carry_valid = false;
    u32 rem = 0;
    // strictly increasing, every neighbouring pair of the file once (also across slices); cross: this lane's pair is not a
    // re-read one (its predecessor is the record in front of it)
    auto order = [&](u64 x0, u64 x1, bool cross) {
        const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)run_carry, (int)(u32)x1, 0x138, 0xF, 0xF, false);          // wave_shr:1
        const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(run_carry >> 32), (int)(u32)(x1 >> 32), 0x138, 0xF, 0xF, false);
        const u64 prev = ((u64)hi << 32) | lo;
        if (cross && (lane > 0 || carry_valid)) {
            if (prev == x0) flags |= PF_FLAG_DUP;
            if (prev > x0) flags |= PF_FLAG_UNSORTED;
        }
        if (x0 == x1) flags |= PF_FLAG_DUP;
        if (x0 > x1) flags |= PF_FLAG_UNSORTED;
        run_carry = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x1 >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x1, 63);
        carry_valid = true;
    };
    auto general_step = [&](u32 lo, bool first, bool dead) {
        const u32 cnt = rem < 128u ? rem : 128u;
        // one record: at the start of its file the pair (0, 1) -- the file has two records --, else the pair (-1, 0)
        const u32 back = (cnt == 1u && !first) ? 1u : 0u;
        const u32 slo = back ? 1u : lo, shi = cnt + back;
        const u32 pmax = shi > 2u ? shi - 2u : 0u;
        const u32 i0 = l2 < pmax ? l2 : pmax;  // the records this lane holds: i0, i0 + 1
        const u32 ri = 1u - back + i0;         // (record index from ptr - 8 bytes)
        const pf_pair pr = *(const pf_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)(ptr - 8) + 8u * ri);
        pf_tpair tq = pf_tpair{0, 0};
        if (FOLD_TAX && tptr && !dead)  // (wave-uniform test)
            tq = *(const pf_tpair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)(tptr - 4) + 4u * ri);
        // (a moved-back pair begins with the previous step's last record itself: nothing in front of it to compare with)
        order(pr.x, pr.y, l2 <= pmax && !back);
        if (!dead) {
            const u64 x[2] = {pr.x, pr.y};
            const u32 tb[2] = {tq.x, tq.y};
            // every record of the step exactly ONCE (hits are counted): the lanes whose own pair fits, and -- an odd number
            // of records -- the first lane behind them for the last record (the others re-read that pair: not theirs)
            const bool own = l2 <= pmax;
            const bool v[2] = {own && i0 - slo < shi - slo, (own || l2 == pmax + 1u) && i0 + 1u - slo < shi - slo};
            process(std::integral_constant<int, 2>{}, x, tb, v);
        }
        rem -= cnt;
        ptr += 1024;
        if (tptr) tptr += 512;
    };
    constexpr int PFU = 2;
    auto full_batch = [&](bool dead) {
        pf_pair pr[PFU];
        pf_tpair tq[PFU];
#pragma unroll
        for (int u = 0; u < PFU; u++) {
            pr[u] = *(const pf_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)ptr + (16u * (u32)lane + 1024u * (u32)u));
            tq[u] = pf_tpair{0, 0};
            if (FOLD_TAX && tptr && !dead)
                tq[u] = *(const pf_tpair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)tptr + (8u * (u32)lane + 512u * (u32)u));
        }
        u64 x[2 * PFU];
        u32 tb[2 * PFU];
        bool v[2 * PFU];
#pragma unroll
        for (int u = 0; u < PFU; u++) {
            order(pr[u].x, pr[u].y, true);
            x[2 * u] = pr[u].x;
            x[2 * u + 1] = pr[u].y;
            tb[2 * u] = tq[u].x;
            tb[2 * u + 1] = tq[u].y;
            v[2 * u] = v[2 * u + 1] = true;
        }
        if (!dead) process(std::integral_constant<int, 2 * PFU>{}, x, tb, v);
        rem -= 128u * PFU;
        ptr += 1024ull * PFU;
        if (tptr) tptr += 512ull * PFU;
    };
    auto take = [&]() -> u32 {
        u32 j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1u);
        return (u32)__builtin_amdgcn_readfirstlane((int)j);
    };
    struct Meta { u64 beg, end, len, f, t; };
    auto fetch = [&](u32 j) -> Meta {  // j = 0 .. S1 - 1 stands for file j + 1
        Meta m = {0, 0, 0, 0, 0};
        if (j < S1) {
            m.beg = sload_u64(&a.cuts[(u64)r * S1 + j]);
            m.end = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]);
            m.f = sload_u64(&a.tab[j + 1]);
            m.t = sload_u64(&a.tab[(u64)S + j + 1]);
            m.len = sload_u64(&a.tab[2 * (u64)S + j + 1]);
        }
        return m;
    };
    // The reference stops reading when its result is empty (inter.go:268-278, diff.go:441-452).  Here every file is still
    // read and checked for its order -- an unsorted or duplicated file must send the call to the exact routes whatever the
    // result -- but a range none of whose records can survive any more stops PROBING: its later slices are streamed with
    // the order check alone (no taxid loads, no table, no LDS).  A wave looks every eighth slice: inter: no record has a
    // hit from every file finished so far; diff: every record has been hit.
    u32 slices = 0;
    auto range_is_dead = [&]() -> bool {
        const u32 finished = OP == UKM_OP_INTER ? __hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
        bool any = false;
        for (u32 i = (u32)lane; i < ne; i += 64) {
            const u32 w = CNT(i);
            any |= OP == UKM_OP_INTER ? ((EULER ? (w & PF_CNT_MASK) : w) >= finished) : w == 0;
        }
        return __ballot(any) == 0ull;
    };
    u32 j = take();
    Meta cur = fetch(j);
    while (j < S1) {
        const u32 jn = take();
        const Meta nxt = fetch(jn);
        const auto f = as_global((const u64 *)(uintptr_t)cur.f);
        const auto t = as_global((const u32 *)(uintptr_t)cur.t);
        const u64 len = cur.len, end = cur.end < cur.beg ? cur.beg : cur.end;
        bool dead = __builtin_amdgcn_readfirstlane((int)s_dead) != 0;
        if (!dead && (++slices & 7u) == 0 && range_is_dead()) {
            dead = true;
            if (lane == 0) s_dead = 1;
        }
        const u64 n = end - cur.beg;
        if (n >= 0xFFFFFF00ull) flags |= PF_FLAG_DUP;  // (a slice of 2^32 records: the exact routes)
        else if (len < 2) {  // (a one-record file: no 16-byte load fits)
            if (!dead && n && lane == 0) {
                const int i0 = find(f[0]);
                if (i0 >= 0) hit(i0, (FOLD_TAX && t) ? t[0] : 0u);
            }
        } else if (n) {
            const u32 lo = cur.beg ? 1u : 0u;
            ptr = cur.f + 8ull * (cur.beg - lo);
            tptr = cur.t ? cur.t + 4ull * (cur.beg - lo) : 0ull;
            rem = (u32)n + lo;
            carry_valid = false;
            general_step(lo, true, dead);
            while (rem >= 128u * PFU) full_batch(dead);
            while (rem) general_step(0u, false, dead);
        }
        // RELEASE: the wave's hit atomics on s_cnt are ordered before the count of finished files that the `need` shortcut
        // reads with ACQUIRE (the hardware keeps a wave's LDS operations in order; the release keeps the compiler from
        // moving the increment in front of them)
        if (OP == UKM_OP_INTER && lane == 0) (void)__hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        j = jn;
        cur = nxt;
    }
    if (flags) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)flags);
    __syncthreads();
    // survivors in file-0 order: ordered compactions of 512 records each
    u32 base = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)tid + (u32)k * NT;
        bool alive = false;
        u32 w = 0;
        if (i < ne) {
            w = CNT(i);
            alive = OP == UKM_OP_INTER ? (EULER ? (w & PF_CNT_MASK) : w) == S1 : w == 0;
        }
        u32 total = 0;
        const u32 excl = block_excl_scan_u32<NT>(alive ? 1u : 0u, s_scan, &total);
        if (alive) {
            const u64 o = (u64)r * L + base + excl;
            a.tmp_k[o] = ent[k];
            if (TAX) {
                u32 tx = TAXR(i);
                if (EULER && (w & PF_NEQ)) {
                    const u32 mn = s_st4[EULER ? i : 0].z, mx = s_st4[EULER ? i : 0].w;
                    if (w & PF_BAD) tx = 0u;
                    else if (a.T.clade8 == nullptr) tx = lca_dev(a.T, a.T.node_at[mn], a.T.node_at[mx]);
                    else if ((mn >> 24) != (mx >> 24)) tx = lca_clade_pair(a.T, mn >> 24, mx >> 24);
                    else tx = lca_dev(a.T, a.T.node_at[mn & 0xFFFFFFu], a.T.node_at[mx & 0xFFFFFFu]);
                }
                a.tmp_t[o] = tx;
            }
        }
        base += total;
    }
    if (tid == 0) a.cnt[r] = base;
}

// ranges -> contiguous output: workgroup r copies its cnt[r] survivors to out[excl[r] ...)
__global__ void pf_gather_kernel(const u64 *tmp_k, const u32 *tmp_t, const u64 *cnt, const u64 *excl, u64 *out, u32 *tout,
                                 u64 out_cap, u32 L) {
    const u32 r = blockIdx.x;
    const u64 n = cnt[r], base = excl[r];
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 pos = base + i;
        if (pos < out_cap) {
            out[pos] = tmp_k[(size_t)r * L + i];
            if (tout) tout[pos] = tmp_t ? tmp_t[(size_t)r * L + i] : 0u;
        }
    }
}

}  // namespace

bool ukm_pfold_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_NO_PFOLD", '1'); }

int ukm_dev_probe_fold(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                       u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback) {
    *fallback = true;
    *n_out = 0;
    if (op != UKM_OP_INTER && op != UKM_OP_DIFF) return UKM_OK;
    if (op == UKM_OP_INTER && (flags & UKM_F_MIX_TAXID)) return UKM_OK;
    const bool cmp = op == UKM_OP_DIFF && tax && (flags & UKM_F_CMP_TAXID);
    {
        const char *e = ukm_env(c, "UKM_PFOLD_TAX");  // developer knob: 0 = inter with taxids through the range fold of ukm_fold.hip
        if (op == UKM_OP_INTER && tax && e && e[0] == '0') return UKM_OK;
    }
    if (S < 2 || lens[0] == 0) return UKM_OK;
    for (int j = 0; j < S; j++)
        if (lens[j] == 0) return UKM_OK;  // (the caller drops / truncates at empty files; anything else: not here)
    if (tax && (op == UKM_OP_INTER || cmp) && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "probe fold: taxids given but out_taxids is NULL");
    // one round of resident workgroups when the first file allows it
    const bool euler = op == UKM_OP_INTER && tax;
    static std::atomic<int> slots_cache[2];
    if (!slots_cache[euler].load(std::memory_order_relaxed)) {
        int per_cu = 0;
        const hipError_t e = euler ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pf_probe_kernel<UKM_OP_INTER, true>, PF_NT_EULER, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pf_probe_kernel<UKM_OP_DIFF, true, true>, PF_NT, 0);
        if (e != hipSuccess || per_cu <= 0) per_cu = 2;
        slots_cache[euler].store(per_cu * c->num_cu, std::memory_order_relaxed);
    }
    const u64 slots = (u64)slots_cache[euler].load(std::memory_order_relaxed);
    const u64 maxl = euler ? 2048ull : (u64)PF_NT * PF_PER;
    u64 L = (lens[0] + slots - 1) / slots;
    L = std::min<u64>(std::max<u64>(L, PF_MINL), maxl);
    const u64 R64 = (lens[0] + L - 1) / L;
    if (R64 > 0x7FFFFFFEull) return UKM_OK;
    {
        // (the shape guard of ukm_fold.hip: a tiny first file against huge later ones leaves whole files to a few CUs)
        u64 rest = 0;
        for (int j = 1; j < S; j++) rest += lens[j];
        if (rest / (u64)(S - 1) / R64 > 16ull * PF_MAXL) return UKM_OK;
    }
    const size_t ntab = (size_t)3 * S;
    std::vector<u64> tab(ntab);
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        tab[(size_t)2 * S + j] = lens[j];
    }
    u64 *d_tab = nullptr;
    UKM_TRY(ws_alloc_t(c, ntab, &d_tab));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), ntab * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame

    PfArgs a;
    memset(&a, 0, sizeof(a));
    a.tab = d_tab;
    a.S = (u32)S;
    a.R = (u32)R64;
    a.L = (u32)L;
    a.T = ukm_taxdev(c);
    u64 *excl = nullptr;
    UKM_TRY(ws_alloc_t(c, ((size_t)a.R + 1) * (S - 1), &a.cuts));
    UKM_TRY(ws_alloc_t(c, (size_t)a.R * L, &a.tmp_k));
    if (tax) UKM_TRY(ws_alloc_t(c, (size_t)a.R * L, &a.tmp_t));
    UKM_TRY(ws_alloc_t(c, (size_t)a.R, &a.cnt));
    UKM_TRY(ws_alloc_t(c, (size_t)a.R + 1, &excl));
    UKM_TRY(ws_alloc_t(c, 8, &a.ctl));
    UKM_HIP(hipMemsetAsync(a.ctl, 0, 8 * sizeof(u64), c->stream));
    const u64 ncuts = ((u64)a.R + 1) * (u64)(S - 1);
    hipLaunchKernelGGL(pf_cuts_kernel, dim3((unsigned)((ncuts + 255) / 256)), dim3(256), 0, c->stream, a);
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (op == UKM_OP_INTER) {
        if (tax) hipLaunchKernelGGL((pf_probe_kernel<UKM_OP_INTER, true>), dim3(a.R), dim3(PF_NT_EULER), 0, c->stream, a);
        else hipLaunchKernelGGL((pf_probe_kernel<UKM_OP_INTER, false>), dim3(a.R), dim3(PF_NT), 0, c->stream, a);
    } else {
        if (cmp) hipLaunchKernelGGL((pf_probe_kernel<UKM_OP_DIFF, true, true>), dim3(a.R), dim3(PF_NT), 0, c->stream, a);
        else if (tax) hipLaunchKernelGGL((pf_probe_kernel<UKM_OP_DIFF, true>), dim3(a.R), dim3(PF_NT), 0, c->stream, a);
        else hipLaunchKernelGGL((pf_probe_kernel<UKM_OP_DIFF, false>), dim3(a.R), dim3(PF_NT), 0, c->stream, a);
    }
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    UKM_TRY(ukm_dev_exclusive_scan_u64(c, a.cnt, excl, a.R, a.ctl));  // ctl[0] = total
    hipLaunchKernelGGL(pf_gather_kernel, dim3(a.R), dim3(256), 0, c->stream, a.tmp_k, tax ? a.tmp_t : nullptr, a.cnt, excl, out,
                       tax ? tout : nullptr, out_cap, a.L);
    UKM_HIP(hipGetLastError());
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, a.ctl, h, 2));
    if (ukm_env(c, "UKM_FOLD_DEBUG"))
        fprintf(stderr, "[pfold] op=%d S=%d R=%u L=%u slots=%llu tax=%d flags=%llu out=%llu\n", op, S, a.R, a.L, (unsigned long long)slots,
                (int)tax, (unsigned long long)h[1], (unsigned long long)h[0]);
    if (h[1] != 0) return UKM_OK;  // duplicate / unsorted / empty marker: the routes behind this one handle and report it
    *n_out = h[0];
    if (h[0] > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu records, capacity is %llu", (unsigned long long)h[0], (unsigned long long)out_cap);
    *fallback = false;
    return UKM_OK;
}
