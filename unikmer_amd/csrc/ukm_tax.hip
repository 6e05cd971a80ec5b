// ukm_tax.hip — taxonomy on the device: replaces bio/taxdump's NewTaxonomyFromNCBI /
// LoadMergedNodesFromNCBI / LCA as used by the reference (util.go:119-171; taxondb.LCA call
// sites union.go:199, inter.go:235,238, diff.go:362,407, common.go:265, sort.go:491,515,
// util-sort.go:128,151,325,374).
//
// Layout in HBM: dense parent[taxid] (u32), depth[taxid] (u8), merged[taxid] (u32, when merged.dmp is given) and the
// root-path table anc[chunk][taxid] (uint4: the ancestors at depths 4 chunk .. 4 chunk + 3) built from them on the
// device.  LCA = resolve merged ids, then the last equal entry of the two root paths (ukm_device.h: two independent
// 16-byte reads for pairs that diverge within four levels of the root; rounds 1-2 climbed parent[] in lock step).
// Contract (taxdump parity is unpinned): see include/unikmer_hip.h.
#include <cstdlib>
#include <vector>

#include "ukm_device.h"

namespace {

__global__ void lca_bulk_kernel(TaxDev T, const u32 *a, const u32 *b, u64 n, u32 *out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lca_dev(T, a[i], b[i]);
}

// root paths: thread t climbs from t to its root once and writes the node it passes at depth l into slot l of t's row
__global__ void build_anc_kernel(const u32 *parent, const u8 *depth, u32 size, u32 *anc) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= size || parent[t] == 0) return;
    u32 x = t;
    for (int l = depth[t]; l >= 0; l--) {
        anc[((size_t)(l >> 2) * size + t) * 4 + (l & 3)] = x;
        x = parent[x];
    }
}

// Decisions over FILE taxids (one per stream, round 5), by ONE thread with the device's own lca_dev so that they agree
// with what the kernels would work out record by record:
//   plan[0]     the left fold LCA(LCA(t0, t1), t2) ... of `inter` / `common` (inter.go:229-239; mix: a zero on either
//               side yields the other)
//   plan[2 + j] diff -t: file j takes no matched code away from file 0 (its taxid equals file 0's or lies below it,
//               diff.go:404-409)
__global__ __launch_bounds__(256) void ct_plan_kernel(TaxDev T, const u32 *ct, u32 n, u32 mix, u32 *plan) {
    __shared__ u32 s_min, s_max, s_zero, s_diff;
    const u32 tid = threadIdx.x;
    const u32 t0 = n ? ct[0] : 0u;
    if (tid == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; s_zero = 0u; s_diff = 0u; }
    __syncthreads();
    // (1000 files: a thread per file -- one thread walking them is a chain of a thousand dependent table reads)
    for (u32 j = tid; j < n; j += 256) {
        const u32 t = ct[j];
        plan[2 + j] = (t0 == t || lca_dev(T, t, t0) == t0) ? 1u : 0u;
        const u32 e = (T.euler && t < T.size) ? T.euler[t] : 0u;
        atomicMin(&s_min, e);
        atomicMax(&s_max, e);
        if (e == 0) s_zero = 1u;
        if (t != t0) s_diff = 1u;
    }
    __syncthreads();
    if (tid != 0) return;
    u32 acc = t0;
    if (mix) {
        // (a zero on either side yields the other -- and the LCA of two trees is a zero: the rule depends on the order)
        for (u32 j = 1; j < n; j++) {
            const u32 b = ct[j];
            acc = acc == 0 ? b : (b == 0 ? acc : lca_dev(T, acc, b));
        }
    } else if (s_diff) {
        // the left fold of lca_dev over a set: 0 when one member has no pre-order number (taxid 0, unknown ids), else the LCA
        // of the members with the smallest and the largest number (the argument of ukm_pfold.hip)
        acc = s_zero ? 0u : lca_dev(T, T.node_at[s_min], T.node_at[s_max]);
    }
    plan[0] = acc;
    plan[1] = 0;
}

// copy a (host or device) array to a host vector
template <typename T>
int to_host(ukm_ctx *c, const T *p, u64 n, std::vector<T> &v) {
    v.resize(n);
    if (n == 0) return UKM_OK;
    if (ukm_is_device_ptr(p)) {
        UKM_HIP(hipMemcpyAsync(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        UKM_HIP(hipStreamSynchronize(c->stream));
    } else {
        memcpy(v.data(), p, n * sizeof(T));
    }
    return UKM_OK;
}

}  // namespace

// device bytes of the taxonomy a context holds now (credited when it is about to be replaced)
static u64 tax_bytes_held(const ukm_ctx *c) {
    if (!c->tax_parent) return 0;
    return (u64)c->tax_size * (sizeof(u32) + sizeof(u8) + (c->tax_merged ? sizeof(u32) : 0) + 2 * sizeof(u32) + (c->tax_clade ? 2 : (c->tax_clade8 ? 1 : 0))) +
           (u64)c->tax_nchunks * c->tax_size * sizeof(uint4) + (c->tax_pair ? ((u64)c->tax_kp * c->tax_kp + 4 + 5 * TAX_CPATH_ROWS) * sizeof(u32) : 0) +
           (c->tax_top ? (u64)c->tax_top_n * sizeof(uint4) : 0);
}

extern "C" int ukm_taxonomy_load(ukm_ctx *c, const uint32_t *child, const uint32_t *parent, uint64_t n,
                                 const uint32_t *merged_old, const uint32_t *merged_new, uint64_t m) {
    if (!c || !child || !parent || n == 0) UKM_FAIL(UKM_ERR_INVALID, "ukm_taxonomy_load: bad argument");
    if (m && (!merged_old || !merged_new)) UKM_FAIL(UKM_ERR_INVALID, "ukm_taxonomy_load: merged arrays missing");
    UKM_HIP(hipSetDevice(c->device));
    std::vector<u32> ch, pa, mo, mn;
    UKM_TRY(to_host(c, child, n, ch));
    UKM_TRY(to_host(c, parent, n, pa));
    UKM_TRY(to_host(c, merged_old, m, mo));
    UKM_TRY(to_host(c, merged_new, m, mn));
    u32 mx = 0, mx_node = 0;
    for (u64 i = 0; i < n; i++) {
        if (ch[i] == 0 || pa[i] == 0) UKM_FAIL(UKM_ERR_INVALID, "ukm_taxonomy_load: taxid 0 is reserved");
        mx = std::max(mx, std::max(ch[i], pa[i]));
        mx_node = std::max(mx_node, ch[i]);
    }
    for (u64 i = 0; i < m; i++) mx = std::max(mx, std::max(mo[i], mn[i]));
    const u64 size = (u64)mx + 1;
    {
        // the tables are dense in the taxid (at least 27 bytes per id on the device, about as much on the host while they
        // are built): refuse sparse huge ids here, before any of it is allocated
        // (the tables being replaced are credited here as they are below: a reload on a nearly full device must not be
        //  refused by this early guard when the later one, which frees the old tables first, would let it fit)
        size_t free_b = 0, total_b = 0;
        UKM_HIP(hipMemGetInfo(&free_b, &total_b));
        const u64 old_b = tax_bytes_held(c);
        if (size * 27 > ((u64)free_b + old_b) / 10 * 9) {  // (the context's cached workspace counts as free: give it back and look again)
            (void)ukm_ctx_trim(c);
            UKM_HIP(hipMemGetInfo(&free_b, &total_b));
        }
        if (size * 27 > ((u64)free_b + old_b) / 10 * 9)
            UKM_FAIL(UKM_ERR_NOMEM,
                     "ukm_taxonomy_load: dense tables for the largest taxid %u need at least %.2f GB, the device has %.2f GB free "
                     "(the taxonomy being replaced counted); renumber sparse taxids densely",
                     mx, size * 27 / 1e9, ((u64)free_b + old_b) / 1e9);
    }
    std::vector<u32> P(size, 0), M;
    std::vector<u8> D(size, 0);
    for (u64 i = 0; i < n; i++) P[ch[i]] = pa[i];
    // a parent that never appears as a child is treated as a root of its own
    for (u64 i = 0; i < n; i++)
        if (P[pa[i]] == 0) P[pa[i]] = pa[i];
    // depths: iterative walk with memoisation; 0xFF marks "unknown yet"
    std::vector<int> depth(size, -1);
    std::vector<u32> stack;
    for (u64 t = 1; t < size; t++) {
        if (P[t] == 0 || depth[t] >= 0) continue;
        stack.clear();
        u32 x = (u32)t;
        while (depth[x] < 0) {
            if (P[x] == x) { depth[x] = 0; break; }
            stack.push_back(x);
            if (stack.size() > 250) UKM_FAIL(UKM_ERR_INVALID, "ukm_taxonomy_load: tree deeper than 250 or cyclic at taxid %u", x);
            x = P[x];
        }
        int d = depth[x];
        while (!stack.empty()) {
            d++;
            depth[stack.back()] = d;
            stack.pop_back();
        }
    }
    for (u64 t = 0; t < size; t++) D[t] = depth[t] < 0 ? 0 : (u8)depth[t];
    if (m) {
        M.assign(size, 0);
        for (u64 i = 0; i < m; i++) M[mo[i]] = mn[i];
    }
    // Pre-order numbers of the forest (children in increasing taxid order; any order works): the LCA of a set of nodes is
    // the LCA of the members with the smallest and the largest number, which lets a fold over MANY files keep a minimum
    // and a maximum per record (two commutative LDS atomics per hit) and do ONE table LCA at the end (ukm_pfold.hip).
    std::vector<u32> E(size, 0), N(1, 0);  // N[0] unused: numbers start at 1
    {
        std::vector<u32> first(size + 1, 0), kids;  // CSR of the children lists
        for (u64 t = 1; t < size; t++)
            if (P[t] != 0 && P[t] != t) first[P[t] + 1]++;
        for (u64 t = 0; t < size; t++) first[t + 1] += first[t];
        kids.resize(first[size]);
        {
            std::vector<u32> fill(first.begin(), first.end() - 1);
            for (u64 t = 1; t < size; t++)
                if (P[t] != 0 && P[t] != t) kids[fill[P[t]]++] = (u32)t;
        }
        std::vector<std::pair<u32, u32>> st;   // (node, next child slot)
        for (u64 r = 1; r < size; r++) {
            if (P[r] != r) continue;
            st.clear();
            st.emplace_back((u32)r, first[r]);
            E[r] = (u32)N.size();
            N.push_back((u32)r);
            while (!st.empty()) {
                auto &top = st.back();
                if (top.second < first[top.first + 1]) {
                    const u32 k = kids[top.second++];
                    E[k] = (u32)N.size();
                    N.push_back(k);
                    st.emplace_back(k, first[k]);
                } else {
                    st.pop_back();
                }
            }
        }
        if (m)
            for (u64 t = 1; t < size; t++)
                if (P[t] == 0 && M[t] != 0 && M[t] < size && P[M[t]] != 0) E[t] = E[M[t]];  // (lca_dev resolves exactly these)
    }
    // Clade codes (TaxDev::clade / top): D = the deepest level <= 3 whose nodes of depth <= D fit 16 bits; every present
    // id gets the index of its ancestor at depth min(own depth, D) -- the pre-order list N hands parents out before their
    // children.  A forest with more than 65535 roots gets no table (Q stays empty: lca_dev goes to the root paths).
    std::vector<unsigned short> Q;
    std::vector<unsigned char> Q8;
    std::vector<u32> PAIR;
    bool one_byte = false;
    std::vector<uint4> T4(1, make_uint4(0, 0, 0, 0));
    {
        u64 by_depth[4] = {0, 0, 0, 0};
        for (size_t i = 1; i < N.size(); i++)
            if (D[N[i]] < 4) by_depth[D[N[i]]]++;
        int Dq = -1;
        u64 acc = 0;
        for (int d = 0; d < 4; d++) {
            acc += by_depth[d];
            if (acc <= 65535) Dq = d; else break;
        }
        // ONE byte per id (the form the kernels prefer: a table a quarter the size of a 4-byte column stays in L2, and the
        // folds keep code << 24 | number in one word): up to 255 clade nodes chosen as a CUT of the forest that follows its
        // shape instead of one depth -- start from the roots and keep replacing the clade node with the most ids below it by
        // ALL of its children while the total fits (NCBI: the cut runs through the phyla and classes of the large kingdoms
        // and stays high in the small ones; the 8-ary tree: the 73 nodes of depth <= 2 and 22 of them split once more).  What
        // the kernels rely on: the set is closed upwards and a clade node has either all of its children in it or none, so
        // (1) two ids with different codes have the LCA of their clade nodes, (2) codes handed out in pre-order are monotone
        // in the pre-order numbers.
        u64 nroots = 0;
        for (size_t i2 = 1; i2 < N.size(); i2++) nroots += P[N[i2]] == N[i2] ? 1u : 0u;
        if (nroots >= 1 && nroots <= 255 && N.size() < (1u << 24)) {
            one_byte = true;
            std::vector<u32> below(size, 0), nchild(size, 0);
            for (size_t i2 = N.size() - 1; i2 >= 1; i2--) {  // (reverse pre-order: children in front of their parents)
                const u32 t = N[i2];
                below[t] += 1;
                if (P[t] != t) { below[P[t]] += below[t]; nchild[P[t]]++; }
            }
            std::vector<char> expanded(size, 0);
            std::vector<std::pair<u32, u32>> heap;  // (ids below, node)
            u64 total = nroots;
            for (size_t i2 = 1; i2 < N.size(); i2++)
                if (P[N[i2]] == N[i2] && nchild[N[i2]]) heap.emplace_back(below[N[i2]], N[i2]);
            std::make_heap(heap.begin(), heap.end());
            // (children of the node that was split join the heap; a node whose children do not fit any more is dropped)
            std::vector<u32> first2(size + 1, 0), kids2;
            for (u64 t = 1; t < size; t++)
                if (P[t] != 0 && P[t] != t) first2[P[t] + 1]++;
            for (u64 t = 0; t < size; t++) first2[t + 1] += first2[t];
            kids2.resize(first2[size]);
            {
                std::vector<u32> fill(first2.begin(), first2.end() - 1);
                for (u64 t = 1; t < size; t++)
                    if (P[t] != 0 && P[t] != t) kids2[fill[P[t]]++] = (u32)t;
            }
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const u32 node = heap.back().second;
                heap.pop_back();
                if (total + nchild[node] > 255) continue;
                expanded[node] = 1;
                total += nchild[node];
                for (u32 q = first2[node]; q < first2[node + 1]; q++) {
                    const u32 k = kids2[q];
                    if (nchild[k]) { heap.emplace_back(below[k], k); std::push_heap(heap.begin(), heap.end()); }
                }
            }
            Q.assign(size, 0);
            std::vector<u32> fnode(1, 0);  // clade node of a code
            for (size_t i2 = 1; i2 < N.size(); i2++) {
                const u32 t = N[i2];
                if (P[t] == t || expanded[P[t]]) {
                    Q[t] = (unsigned short)fnode.size();
                    fnode.push_back(t);
                } else {
                    Q[t] = Q[P[t]];
                }
            }
            if (m)
                for (u64 t = 1; t < size; t++)
                    if (P[t] == 0 && M[t] != 0 && M[t] < size && P[M[t]] != 0) Q[t] = Q[M[t]];
            Q8.assign(Q.begin(), Q.end());
            // the LCA of every pair of clade nodes (at most 256 x 256), by the parent / depth climb
            const u32 kp = (u32)fnode.size();
            T4.assign(kp, make_uint4(0, 0, 0, 0));  // (the two-byte form's rows: not used beside the pair table)
            PAIR.assign((size_t)kp * kp, 0u);
            for (u32 i2 = 1; i2 < kp; i2++)
                for (u32 j2 = 1; j2 < kp; j2++) {
                    if (i2 == j2) continue;
                    u32 a = fnode[i2], b = fnode[j2];
                    while (depth[a] > depth[b]) a = P[a];
                    while (depth[b] > depth[a]) b = P[b];
                    while (a != b && P[a] != a && P[b] != b) { a = P[a]; b = P[b]; }
                    PAIR[(size_t)i2 * kp + j2] = a == b ? a : 0u;  // (different trees: 0)
                }
            // behind it (TaxDev::cpath / cnode): the first 16 levels of every clade node's root path as code bytes, and the
            // nodes themselves -- 5 KB a workgroup keeps in LDS.  The set is closed upwards, so a clade node's depth in the
            // forest is its depth among the clade nodes and every ancestor has a code.
            PAIR.resize((PAIR.size() + 3) & ~(size_t)3, 0u);
            const size_t off = PAIR.size();
            PAIR.resize(off + 5 * (size_t)TAX_CPATH_ROWS, 0u);
            unsigned char *rows = reinterpret_cast<unsigned char *>(PAIR.data() + off);
            for (u32 i2 = 1; i2 < kp; i2++) {
                PAIR[off + 4 * (size_t)TAX_CPATH_ROWS + i2] = fnode[i2];
                for (u32 t = fnode[i2];; t = P[t]) {
                    if (depth[t] < 16) rows[(size_t)i2 * 16 + depth[t]] = (unsigned char)Q[t];
                    if (P[t] == t) break;
                }
            }
        } else if (Dq >= 0) {
            Q.assign(size, 0);
            for (size_t i = 1; i < N.size(); i++) {
                const u32 t = N[i];
                if ((int)D[t] <= Dq) {
                    uint4 row = D[t] == 0 ? make_uint4(0, 0, 0, 0) : T4[Q[P[t]]];
                    (D[t] == 0 ? row.x : D[t] == 1 ? row.y : D[t] == 2 ? row.z : row.w) = t;
                    Q[t] = (unsigned short)T4.size();
                    T4.push_back(row);
                } else {
                    Q[t] = Q[P[t]];
                }
            }
            // a merged id has its target's code (and its target's pre-order number, above): pairs with an unrelated id are
            // settled by the codes, anything closer resolves the id on the way to the root paths
            if (m)
                for (u64 t = 1; t < size; t++)
                    if (P[t] == 0 && M[t] != 0 && M[t] < size && P[M[t]] != 0) Q[t] = Q[M[t]];
        }
    }
    // ---- device tables: ALL of them are built beside the context's current ones and swapped in only when every
    // allocation, copy and kernel has succeeded; a failed load leaves the context exactly as it was (a half-replaced
    // taxonomy would pass the `tax_parent != nullptr` guards and send the kernels through a null root-path table).
    // Root-path table for lca_dev: 16 bytes per id per 4 levels (NCBI: ~3.4 M dense ids x ~45 levels = 0.65 GB of the
    // 288 GB; the first chunk -- all that random pairs touch -- is 54 MB).  The tables are DENSE in the taxid: a dump with
    // sparse huge ids (hash-like ids up to 2^31) multiplies that by the largest id, so the load is refused with a clear
    // message when the tables would not fit the device's free memory instead of failing half-way through.
    int maxd = 0;
    for (u64 t = 0; t < size; t++) maxd = std::max(maxd, (int)D[t]);
    const u32 nchunks = (u32)(maxd + 1 + 3) / 4;
    const u64 anc_bytes = (u64)nchunks * size * sizeof(uint4);
    const u64 need = size * (sizeof(u32) + sizeof(u8) + (m ? sizeof(u32) : 0) + sizeof(u32) + sizeof(unsigned short)) + N.size() * sizeof(u32) +
                     T4.size() * sizeof(uint4) + anc_bytes;
    {
        // free memory as the driver sees it -- plus what this context can give back: its cached workspace (kept at the
        // high-water mark of the largest call, up to 160 GB) is released before the load is refused, and the taxonomy that
        // is being replaced is credited (it is freed as soon as the new tables stand)
        size_t free_b = 0, total_b = 0;
        UKM_HIP(hipMemGetInfo(&free_b, &total_b));
        const u64 old_bytes = tax_bytes_held(c);
        if (need > (u64)free_b / 10 * 9) {
            (void)ukm_ctx_trim(c);
            UKM_HIP(hipMemGetInfo(&free_b, &total_b));
        }
        const bool fits_beside = need <= (u64)free_b / 10 * 9;
        if (!fits_beside && need <= ((u64)free_b + old_bytes) / 10 * 9) {
            // only WITHOUT the old tables: they go first (the one case in which a failed load leaves no taxonomy behind)
            UKM_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->tax_parent); (void)hipFree(c->tax_depth);
            if (c->tax_merged) (void)hipFree(c->tax_merged);
            (void)hipFree(c->tax_anc); (void)hipFree(c->tax_euler); (void)hipFree(c->tax_node_at);
            if (c->tax_clade) (void)hipFree(c->tax_clade);
            if (c->tax_top) (void)hipFree(c->tax_top);
            if (c->tax_clade8) (void)hipFree(c->tax_clade8);
            if (c->tax_pair) (void)hipFree(c->tax_pair);
            c->tax_clade = nullptr; c->tax_top = nullptr; c->tax_clade8 = nullptr; c->tax_pair = nullptr; c->tax_kp = 0; c->tax_top_n = 0;
            c->tax_parent = nullptr; c->tax_depth = nullptr; c->tax_merged = nullptr; c->tax_anc = nullptr;
            c->tax_euler = nullptr; c->tax_node_at = nullptr; c->tax_size = 0; c->tax_nchunks = 0; c->tax_max = 0;
            UKM_HIP(hipMemGetInfo(&free_b, &total_b));
        }
        if (need > (u64)free_b / 10 * 9)
            UKM_FAIL(UKM_ERR_NOMEM,
                     "ukm_taxonomy_load: the dense tables need %.2f GB (largest taxid %u, depth %d: 16 B per id per 4 levels), the "
                     "device has %.2f GB free; renumber sparse taxids densely",
                     need / 1e9, mx, maxd, free_b / 1e9);
    }
    UKM_HIP(hipStreamSynchronize(c->stream));
    u32 *n_parent = nullptr, *n_merged = nullptr, *n_euler = nullptr, *n_node_at = nullptr;
    u8 *n_depth = nullptr;
    uint4 *n_anc = nullptr, *n_top = nullptr;
    unsigned short *n_clade = nullptr;
    unsigned char *n_clade8 = nullptr;
    u32 *n_pair = nullptr;
    auto drop_new = [&]() {
        if (n_pair) (void)hipFree(n_pair);
        if (n_clade) (void)hipFree(n_clade);
        if (n_clade8) (void)hipFree(n_clade8);
        if (n_top) (void)hipFree(n_top);
        if (n_parent) (void)hipFree(n_parent);
        if (n_depth) (void)hipFree(n_depth);
        if (n_merged) (void)hipFree(n_merged);
        if (n_euler) (void)hipFree(n_euler);
        if (n_node_at) (void)hipFree(n_node_at);
        if (n_anc) (void)hipFree(n_anc);
    };
    int rc = [&]() -> int {
        UKM_HIP(hipMalloc((void **)&n_parent, size * sizeof(u32)));
        UKM_HIP(hipMalloc((void **)&n_depth, size * sizeof(u8)));
        UKM_HIP(hipMemcpy(n_parent, P.data(), size * sizeof(u32), hipMemcpyHostToDevice));
        UKM_HIP(hipMemcpy(n_depth, D.data(), size * sizeof(u8), hipMemcpyHostToDevice));
        if (m) {
            UKM_HIP(hipMalloc((void **)&n_merged, size * sizeof(u32)));
            UKM_HIP(hipMemcpy(n_merged, M.data(), size * sizeof(u32), hipMemcpyHostToDevice));
        }
        UKM_HIP(hipMalloc((void **)&n_euler, size * sizeof(u32)));
        UKM_HIP(hipMalloc((void **)&n_node_at, N.size() * sizeof(u32)));
        UKM_HIP(hipMemcpy(n_euler, E.data(), size * sizeof(u32), hipMemcpyHostToDevice));
        UKM_HIP(hipMemcpy(n_node_at, N.data(), N.size() * sizeof(u32), hipMemcpyHostToDevice));
        if (!Q.empty()) {
            if (one_byte) {
                UKM_HIP(hipMalloc((void **)&n_clade8, size));
                UKM_HIP(hipMemcpy(n_clade8, Q8.data(), size, hipMemcpyHostToDevice));
                UKM_HIP(hipMalloc((void **)&n_pair, PAIR.size() * sizeof(u32)));
                UKM_HIP(hipMemcpy(n_pair, PAIR.data(), PAIR.size() * sizeof(u32), hipMemcpyHostToDevice));
            } else {
                UKM_HIP(hipMalloc((void **)&n_clade, size * sizeof(unsigned short)));
                UKM_HIP(hipMemcpy(n_clade, Q.data(), size * sizeof(unsigned short), hipMemcpyHostToDevice));
            }
            UKM_HIP(hipMalloc((void **)&n_top, T4.size() * sizeof(uint4)));
            UKM_HIP(hipMemcpy(n_top, T4.data(), T4.size() * sizeof(uint4), hipMemcpyHostToDevice));
        }
        UKM_HIP(hipMalloc((void **)&n_anc, anc_bytes));
        UKM_HIP(hipMemsetAsync(n_anc, 0, anc_bytes, c->stream));
        hipLaunchKernelGGL(build_anc_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, c->stream, n_parent, n_depth,
                           (u32)size, (u32 *)n_anc);
        UKM_HIP(hipGetLastError());
        UKM_HIP(hipStreamSynchronize(c->stream));
        return UKM_OK;
    }();
    if (rc != UKM_OK) {
        drop_new();
        return rc;
    }
    // the swap: nothing below can fail
    if (c->tax_parent) (void)hipFree(c->tax_parent);
    if (c->tax_depth) (void)hipFree(c->tax_depth);
    if (c->tax_merged) (void)hipFree(c->tax_merged);
    if (c->tax_anc) (void)hipFree(c->tax_anc);
    if (c->tax_euler) (void)hipFree(c->tax_euler);
    if (c->tax_node_at) (void)hipFree(c->tax_node_at);
    if (c->tax_clade) (void)hipFree(c->tax_clade);
    if (c->tax_top) (void)hipFree(c->tax_top);
    if (c->tax_clade8) (void)hipFree(c->tax_clade8);
    if (c->tax_pair) (void)hipFree(c->tax_pair);
    c->tax_pair = n_pair;
    c->tax_kp = n_pair ? (u32)T4.size() : 0u;
    c->tax_clade = n_clade;
    c->tax_clade8 = n_clade8;
    c->tax_top = n_top;
    c->tax_top_n = n_top ? (u32)T4.size() : 0u;
    c->tax_parent = n_parent;
    c->tax_depth = n_depth;
    c->tax_merged = n_merged;
    c->tax_anc = n_anc;
    c->tax_euler = n_euler;
    c->tax_node_at = n_node_at;
    c->tax_nchunks = nchunks;
    c->tax_size = (u32)size;
    c->tax_max = mx_node;
    return UKM_OK;
}

// *plan: a workspace array of 2 + n words (see ct_plan_kernel), valid until the enclosing top-level call returns
int ukm_dev_ct_plan(ukm_ctx *c, const u32 *ct_host, int n, bool mix, u32 **plan) {
    u32 *d = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)2 * n + 2, &d));
    if (n) {
        UKM_HIP(hipMemcpyAsync(d + 2 + n, ct_host, (size_t)n * sizeof(u32), hipMemcpyHostToDevice, c->stream));
        UKM_HIP(hipStreamSynchronize(c->stream));  // (ct_host is a pageable buffer of the caller's frame)
    }
    hipLaunchKernelGGL(ct_plan_kernel, dim3(1), dim3(256), 0, c->stream, ukm_taxdev(c), d + 2 + n, (u32)n, mix ? 1u : 0u, d);
    UKM_HIP(hipGetLastError());
    *plan = d;
    return UKM_OK;
}

extern "C" int ukm_taxonomy_max_taxid(ukm_ctx *c, uint32_t *max_taxid) {
    if (!c || !max_taxid) UKM_FAIL(UKM_ERR_INVALID, "ukm_taxonomy_max_taxid: NULL argument");
    if (!c->tax_parent) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_taxonomy_max_taxid: no taxonomy loaded");
    *max_taxid = c->tax_max;
    return UKM_OK;
}

extern "C" int ukm_lca(ukm_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out) {
    if (!ctx || (n && (!a || !b || !out))) UKM_FAIL(UKM_ERR_INVALID, "ukm_lca: NULL argument");
    if (!ctx->tax_parent) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_lca: no taxonomy loaded");
    if (n == 0) return UKM_OK;
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u32 *da = nullptr, *db = nullptr;
        u32 *o = nullptr;
        UKM_TRY(ukm_in_t(ctx, a, n, &da));
        UKM_TRY(ukm_in_t(ctx, b, n, &db));
        UKM_TRY(ukm_out_t(ctx, out, n, &o));
        hipLaunchKernelGGL(lca_bulk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           ukm_taxdev(ctx), da, db, n, o);
        UKM_HIP(hipGetLastError());
        return UKM_OK;
    }();
    return ukm_finish(&s, rc);
}
