// ukm_comm.hip — the multi-GPU exchange step of SURVEY.md §8(e) behind the C ABI: one process per GPU,
// the all-to-all-v of contiguous slices of a sorted stream over RCCL / xGMI (grouped ncclSend / ncclRecv).
// The reference has nothing like it (single process, goroutines); this is what a Go host needs to run the
// prefix-sharded path without Python: ukm_prefix_splitters -> ukm_partition_points -> ukm_shard_exchange ->
// ukm_union / ukm_merge_k of the received slices -> the 1-GPU path on the rank's range.
// RCCL is loaded with dlopen on first use, so the library has no link-time dependency on it (a single-GPU
// host never touches it).  unikmer_amd/dist.py is the same protocol over torch.distributed.
#include <dlfcn.h>

#include <vector>

#include "ukm_internal.h"

namespace {

typedef struct { char internal[128]; } UkmNcclId;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *UkmNcclComm;
enum { UKM_NCCL_UINT32 = 3, UKM_NCCL_UINT64 = 5 };  // ncclDataType_t values (rccl.h)

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(UkmNcclId *) = nullptr;
    int (*CommInitRank)(UkmNcclComm *, int, UkmNcclId, int) = nullptr;
    int (*CommDestroy)(UkmNcclComm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, UkmNcclComm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, UkmNcclComm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, UkmNcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.h ? &r : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
    }
    if (!r.h) return nullptr;
#define UKM_SYM(field, name)                                         \
    *(void **)(&r.field) = dlsym(r.h, name);                         \
    if (!r.field) { dlclose(r.h); r.h = nullptr; return nullptr; }
    UKM_SYM(GetUniqueId, "ncclGetUniqueId")
    UKM_SYM(CommInitRank, "ncclCommInitRank")
    UKM_SYM(CommDestroy, "ncclCommDestroy")
    UKM_SYM(AllGather, "ncclAllGather")
    UKM_SYM(Send, "ncclSend")
    UKM_SYM(Recv, "ncclRecv")
    UKM_SYM(GroupStart, "ncclGroupStart")
    UKM_SYM(GroupEnd, "ncclGroupEnd")
    UKM_SYM(GetErrorString, "ncclGetErrorString")
#undef UKM_SYM
    return &r;
}

#define UKM_NCCL(expr)                                                                              \
    do {                                                                                            \
        int _e = (expr);                                                                            \
        if (_e != 0) {                                                                              \
            ukm_set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString(_e), __FILE__, __LINE__); \
            return UKM_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

}  // namespace

extern "C" int ukm_comm_get_unique_id(void *id) {
    if (!id) UKM_FAIL(UKM_ERR_INVALID, "ukm_comm_get_unique_id: id is NULL");
    Rccl *R = rccl();
    if (!R) UKM_FAIL(UKM_ERR_HIP, "ukm_comm_get_unique_id: librccl.so could not be loaded (%s)", dlerror());
    UKM_NCCL(R->GetUniqueId((UkmNcclId *)id));
    return UKM_OK;
}

extern "C" int ukm_comm_init(ukm_ctx *c, int nranks, int rank, const void *id) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) UKM_FAIL(UKM_ERR_INVALID, "ukm_comm_init: bad argument");
    if (c->comm) UKM_FAIL(UKM_ERR_INVALID, "ukm_comm_init: the context already has a communicator");
    Rccl *R = rccl();
    if (!R) UKM_FAIL(UKM_ERR_HIP, "ukm_comm_init: librccl.so could not be loaded (%s)", dlerror());
    UKM_HIP(hipSetDevice(c->device));
    UkmNcclId uid;
    memcpy(&uid, id, sizeof(uid));
    UkmNcclComm comm = nullptr;
    UKM_NCCL(R->CommInitRank(&comm, nranks, uid, rank));
    c->comm = comm;
    c->comm_size = nranks;
    c->comm_rank = rank;
    return UKM_OK;
}

extern "C" int ukm_comm_destroy(ukm_ctx *c) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_comm_destroy: ctx is NULL");
    if (!c->comm) return UKM_OK;
    Rccl *R = rccl();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (R) (void)R->CommDestroy((UkmNcclComm)c->comm);
    c->comm = nullptr;
    c->comm_size = 0;
    c->comm_rank = 0;
    return UKM_OK;
}

extern "C" int ukm_comm_info(ukm_ctx *c, int *nranks, int *rank) {
    if (!c || !nranks || !rank) UKM_FAIL(UKM_ERR_INVALID, "ukm_comm_info: NULL argument");
    *nranks = c->comm ? c->comm_size : 0;
    *rank = c->comm ? c->comm_rank : 0;
    return UKM_OK;
}

// rank g owns [splitters[g], splitters[g + 1]) of the code space [0, 2^key_bits)
extern "C" int ukm_prefix_splitters(int key_bits, int nranks, uint64_t *splitters) {
    if (!splitters || nranks < 1 || key_bits < 1 || key_bits > 64) UKM_FAIL(UKM_ERR_INVALID, "ukm_prefix_splitters: bad argument");
    for (int g = 0; g < nranks; g++) {
        // g * 2^key_bits / nranks without overflow
        const unsigned __int128 top = (unsigned __int128)1 << key_bits;
        splitters[g] = (uint64_t)((top * (unsigned __int128)g) / (unsigned __int128)nranks);
    }
    return UKM_OK;
}

extern "C" int ukm_shard_exchange(ukm_ctx *c, const uint64_t *keys, const uint32_t *taxids, const uint64_t *send_counts,
                                  uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *recv_counts,
                                  uint64_t *n_out) {
    if (!c || !send_counts || !recv_counts || !n_out) UKM_FAIL(UKM_ERR_INVALID, "ukm_shard_exchange: NULL argument");
    if (!c->comm) UKM_FAIL(UKM_ERR_INVALID, "ukm_shard_exchange: ukm_comm_init has not been called on this context");
    Rccl *R = rccl();
    if (!R) UKM_FAIL(UKM_ERR_HIP, "ukm_shard_exchange: librccl.so could not be loaded");
    const int W = c->comm_size, me = c->comm_rank;
    u64 n = 0;
    for (int g = 0; g < W; g++) n += send_counts[g];
    if ((!keys && n) || (taxids && !out_taxids)) UKM_FAIL(UKM_ERR_INVALID, "ukm_shard_exchange: bad argument");
    CallScope s;
    UKM_TRY(ukm_begin(c, &s));
    int rc = [&]() -> int {
        UkmNcclComm comm = (UkmNcclComm)c->comm;
        // 1. everybody learns everybody's slice sizes: all-gather of the W send counts
        u64 *d_cnt = nullptr, *d_all = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)W, &d_cnt));
        UKM_TRY(ws_alloc_t(c, (size_t)W * W, &d_all));
        UKM_HIP(hipMemcpyAsync(d_cnt, send_counts, (size_t)W * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        UKM_NCCL(R->AllGather(d_cnt, d_all, (size_t)W, UKM_NCCL_UINT64, comm, c->stream));
        std::vector<u64> all((size_t)W * W);
        UKM_HIP(hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        UKM_HIP(hipStreamSynchronize(c->stream));
        u64 total = 0;
        for (int g = 0; g < W; g++) {
            recv_counts[g] = all[(size_t)g * W + me];  // what rank g sends to me
            total += recv_counts[g];
        }
        *n_out = total;
        if (total > out_cap)
            UKM_FAIL(UKM_ERR_CAPACITY, "ukm_shard_exchange: %llu records arrive, capacity is %llu", (unsigned long long)total,
                     (unsigned long long)out_cap);
        // 2. the slices themselves: device pointers go to RCCL as they are, host arrays are staged
        const u64 *k = nullptr;
        const u32 *t = nullptr;
        u64 *ok = nullptr;
        u32 *ot = nullptr;
        UKM_TRY(ukm_in_t(c, keys, n, &k));
        if (taxids) UKM_TRY(ukm_in_t(c, taxids, n, &t));
        UKM_TRY(ukm_out_t(c, out_keys, out_cap, &ok));
        if (taxids) UKM_TRY(ukm_out_t(c, out_taxids, out_cap, &ot));
        // one group: every peer's send and receive is posted before any of them blocks (full mesh over xGMI)
        UKM_NCCL(R->GroupStart());
        int first_err = 0;
        u64 so = 0, ro = 0;
        for (int g = 0; g < W && !first_err; g++) {
            if (send_counts[g]) first_err = R->Send(k + so, (size_t)send_counts[g], UKM_NCCL_UINT64, g, comm, c->stream);
            if (!first_err && recv_counts[g]) first_err = R->Recv(ok + ro, (size_t)recv_counts[g], UKM_NCCL_UINT64, g, comm, c->stream);
            if (!first_err && taxids && send_counts[g]) first_err = R->Send(t + so, (size_t)send_counts[g], UKM_NCCL_UINT32, g, comm, c->stream);
            if (!first_err && taxids && recv_counts[g]) first_err = R->Recv(ot + ro, (size_t)recv_counts[g], UKM_NCCL_UINT32, g, comm, c->stream);
            so += send_counts[g];
            ro += recv_counts[g];
        }
        const int end_err = R->GroupEnd();  // always closed, also after a failed post
        if (first_err || end_err)
            UKM_FAIL(UKM_ERR_HIP, "ukm_shard_exchange: RCCL send/recv failed: %s", R->GetErrorString(first_err ? first_err : end_err));
        ukm_out_resize(c, out_keys, total * sizeof(u64));
        if (taxids) ukm_out_resize(c, out_taxids, total * sizeof(u32));
        return UKM_OK;
    }();
    return ukm_finish(&s, rc);
}
