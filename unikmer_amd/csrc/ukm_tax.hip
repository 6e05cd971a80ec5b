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
    UKM_HIP(hipStreamSynchronize(c->stream));
    if (c->tax_parent) { (void)hipFree(c->tax_parent); c->tax_parent = nullptr; }
    if (c->tax_depth) { (void)hipFree(c->tax_depth); c->tax_depth = nullptr; }
    if (c->tax_merged) { (void)hipFree(c->tax_merged); c->tax_merged = nullptr; }
    UKM_HIP(hipMalloc((void **)&c->tax_parent, size * sizeof(u32)));
    UKM_HIP(hipMalloc((void **)&c->tax_depth, size * sizeof(u8)));
    UKM_HIP(hipMemcpy(c->tax_parent, P.data(), size * sizeof(u32), hipMemcpyHostToDevice));
    UKM_HIP(hipMemcpy(c->tax_depth, D.data(), size * sizeof(u8), hipMemcpyHostToDevice));
    if (m) {
        UKM_HIP(hipMalloc((void **)&c->tax_merged, size * sizeof(u32)));
        UKM_HIP(hipMemcpy(c->tax_merged, M.data(), size * sizeof(u32), hipMemcpyHostToDevice));
    }
    c->tax_size = (u32)size;
    c->tax_max = mx_node;
    // root-path table for lca_dev: 16 bytes per node per 4 levels (NCBI: ~3.4 M dense ids x ~45 levels = 0.65 GB of the
    // 288 GB; the first chunk -- all that random pairs touch -- is 54 MB; the deepest tree the loader accepts, 250
    // levels, would take 1 KB per id)
    if (c->tax_anc) { (void)hipFree(c->tax_anc); c->tax_anc = nullptr; c->tax_nchunks = 0; }
    int maxd = 0;
    for (u64 t = 0; t < size; t++) maxd = std::max(maxd, (int)D[t]);
    const u32 nchunks = (u32)(maxd + 1 + 3) / 4;
    const u64 bytes = (u64)nchunks * size * sizeof(uint4);
    UKM_HIP(hipMalloc((void **)&c->tax_anc, bytes));
    UKM_HIP(hipMemsetAsync(c->tax_anc, 0, bytes, c->stream));
    hipLaunchKernelGGL(build_anc_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, c->stream, c->tax_parent,
                       c->tax_depth, (u32)size, (u32 *)c->tax_anc);
    UKM_HIP(hipGetLastError());
    UKM_HIP(hipStreamSynchronize(c->stream));
    c->tax_nchunks = nchunks;
    // Pre-order numbers of the forest (children in increasing taxid order; any order works): the LCA of a set of nodes is
    // the LCA of the members with the smallest and the largest number, which lets a fold over MANY files keep a minimum
    // and a maximum per record (two commutative LDS atomics per hit) and do ONE table LCA at the end (ukm_pfold.hip).
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
        std::vector<u32> E(size, 0), N(1, 0);  // N[0] unused: numbers start at 1
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
        if (c->tax_euler) { (void)hipFree(c->tax_euler); c->tax_euler = nullptr; }
        if (c->tax_node_at) { (void)hipFree(c->tax_node_at); c->tax_node_at = nullptr; }
        UKM_HIP(hipMalloc((void **)&c->tax_euler, size * sizeof(u32)));
        UKM_HIP(hipMalloc((void **)&c->tax_node_at, N.size() * sizeof(u32)));
        UKM_HIP(hipMemcpy(c->tax_euler, E.data(), size * sizeof(u32), hipMemcpyHostToDevice));
        UKM_HIP(hipMemcpy(c->tax_node_at, N.data(), N.size() * sizeof(u32), hipMemcpyHostToDevice));
    }
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
