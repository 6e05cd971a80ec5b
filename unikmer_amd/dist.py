"""Multi-GPU prefix sharding: one process per GPU, torch.distributed over
RCCL ("nccl" backend on ROCm; "gloo" in the CPU tests).

Codes are independent under every set operation, so the VALUE SPACE is partitioned by high
bits: rank g owns [splitter[g], splitter[g+1]).  A sorted stream's share for each rank is a
contiguous slice (cut points come from ukm_partition_points on the GPU), so redistribution is
one all-to-all-v of contiguous slices.  After it every rank runs the single-GPU path on its
range; the global result is the concatenation of the ranks' results in rank order.

Nothing here computes on the CPU: cut points are an argument (the GPU library produces them),
the exchange is pure torch.distributed plumbing.
"""
import torch
import torch.distributed as dist


class _Done:
    """stand-in for an async Work handle of an exchange that has already completed"""

    def wait(self):
        return True


def _all_to_all(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    """dist.all_to_all_single, plus ONE extra case: device tensors over the `gloo` backend (1-GPU boxes where
    several ranks share cuda:0 and the control plane is gloo — tests and bench.py's UKM_BENCH_ONE_GPU hook) are
    staged through host memory, because gloo's all-to-all only takes CPU tensors.  On `nccl` (= RCCL) the
    tensors go to the collective as they are."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(h_out)
        return _Done() if async_op else None
    return dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                                  async_op=async_op)


def prefix_splitters(key_bits, world):
    """world+1 boundaries of the code space [0, 2^key_bits): rank g owns
    [s[g], s[g+1]).  key_bits = 2k for k-mer codes, 64 for hashes."""
    top = 1 << key_bits
    return [min((g * top) // world, (1 << 64) - 1) for g in range(world)] + [top if key_bits < 64 else (1 << 64) - 1]


SPLIT_SAMPLES = 1024   # per rank (the C ABI's UKM_SPLIT_SAMPLES)


def sampled_splitters(files_keys, key_bits, group=None, samples=SPLIT_SAMPLES):
    """world + 1 boundaries cut so that every rank receives about the same number of RECORDS (k-mer
    codes are not uniform in their top bits; equal-width ranges left one of eight ranks with 1.9 x the mean on the
    canonical 31-mers of E. coli).  Collective: every rank passes the sorted files it holds; each contributes `samples`
    values taken at regular positions of its files laid end to end (sample i = position (2 i + 1) n / (2 samples)) and
    its record count; one all-gather; the boundaries come from the library's pure host function
    ukm_shard_splitters_plan, the same arithmetic the C ABI's ukm_shard_splitters runs, so every rank -- and a C or Go
    host -- cuts identically.  Any non-decreasing boundaries give the same concatenated result; these balance it."""
    from . import lib
    world = dist.get_world_size(group)
    live = [k for k in files_keys if k.numel()]
    n = sum(k.numel() for k in live)
    dev = files_keys[0].device if len(files_keys) else torch.device("cpu")
    mine = torch.zeros(samples + 1, dtype=torch.int64, device=dev)
    mine[0] = n
    if n:
        base = [0]
        for k in live:
            base.append(base[-1] + k.numel())
        per_file = [[] for _ in live]
        f = 0
        for i in range(samples):
            ps = ((2 * i + 1) * n) // (2 * samples)
            while ps >= base[f + 1]:
                f += 1
            per_file[f].append(ps - base[f])
        # positions ascend, so taking the files one after the other keeps the samples in position order
        mine[1:] = torch.cat([live[f][torch.tensor(ix, dtype=torch.int64, device=dev)] for f, ix in enumerate(per_file) if ix])
    if dist.get_backend(group) == "gloo":            # CPU control plane (tests, 1-GPU boxes): gather on the host
        hm = mine.cpu()
        gl = [torch.empty_like(hm) for _ in range(world)]
        dist.all_gather(gl, hm, group=group)
        allw = torch.stack(gl)
    else:
        allw = torch.empty((world, samples + 1), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allw, mine, group=group)
        allw = allw.cpu()
    return lib.Context.shard_splitters_plan(world, allw.numpy().view("uint64"), key_bits)


def cuts_to_counts(cuts, n):
    """cut indices (lower bounds of splitters[0..world-1] in the sorted stream) -> slice sizes."""
    c = [int(x) for x in cuts] + [int(n)]
    return [c[i + 1] - c[i] for i in range(len(cuts))]


def exchange_sorted(keys, counts, taxids=None, group=None):
    """All-to-all-v of the contiguous slices of ONE sorted stream.

    keys:   1-D int64 tensor (uint64 bit patterns), sorted, on this rank's device
    counts: list[world] — number of records of `keys` that belong to each rank, in rank order
    returns (recv_keys, recv_taxids, recv_counts): the received slices concatenated in source
    rank order; each slice is sorted, and slices from different ranks cover the SAME value
    range (this rank's), so they are inputs to one n-way merge/union on this rank.
    """
    world = dist.get_world_size(group)
    assert len(counts) == world and sum(counts) == keys.numel()
    send = torch.tensor(counts, dtype=torch.int64, device=keys.device)
    recv = torch.empty(world, dtype=torch.int64, device=keys.device)
    _all_to_all(recv, send, group=group)
    recv_counts = [int(x) for x in recv.cpu()]
    out = torch.empty(sum(recv_counts), dtype=keys.dtype, device=keys.device)
    _all_to_all(out, keys, recv_counts, list(counts), group)
    out_t = None
    if taxids is not None:
        out_t = torch.empty(sum(recv_counts), dtype=taxids.dtype, device=taxids.device)
        _all_to_all(out_t, taxids, recv_counts, list(counts), group)
    return out, out_t, recv_counts


def split_by_counts(t, counts):
    """views of the received buffer, one per source rank"""
    out, off = [], 0
    for c in counts:
        out.append(t[off:off + c])
        off += c
    return out


def _resolve_splitters(files_keys, key_bits, splitters, group):
    """None -> equal-width prefix ranges; "sampled" -> sampled_splitters (collective); a list -> as given"""
    world = dist.get_world_size(group)
    if splitters is None:
        return prefix_splitters(key_bits, world)
    if isinstance(splitters, str):
        assert splitters == "sampled"
        return sampled_splitters(files_keys, key_bits, group)
    assert len(splitters) == world + 1
    return list(splitters)


def _exchange_files(ctx, files_keys, key_bits, files_taxids=None, group=None, splitters=None):
    """Generator over the logical files: cuts every local file at the prefix splitters, ships ALL slice sizes in ONE
    small all-to-all, then yields (pieces, taxid_pieces | None) of file 0, 1, ... -- the sorted slices that arrived for
    this rank's range, one per source rank -- while the all-to-all-v of the NEXT file is already in flight.
    Also returns, through the `info` dict it yields first, the global size of every logical file."""
    world = dist.get_world_size(group)
    spl = _resolve_splitters(files_keys, key_bits, splitters, group)[:-1]
    nfiles = len(files_keys)
    dev = files_keys[0].device if nfiles else None
    counts_all = [cuts_to_counts(ctx.partition_points(k, spl), k.numel()) for k in files_keys]
    recv_counts_all = [[0] * world for _ in range(nfiles)]
    global_sizes = [0] * nfiles
    if nfiles:
        send = torch.tensor(counts_all, dtype=torch.int64, device=dev).t().contiguous()   # [world, nfiles]
        recv = torch.empty_like(send)
        _all_to_all(recv, send, group=group)
        rc = recv.cpu().tolist()                                                            # [source rank][file]
        recv_counts_all = [[int(rc[src][i]) for src in range(world)] for i in range(nfiles)]
        # GLOBAL size of every logical file (`inter` needs it); same small-message pattern
        gs = torch.tensor([k.numel() for k in files_keys], dtype=torch.int64,
                          device=dev if dist.get_backend(group) != "gloo" else "cpu")
        dist.all_reduce(gs, op=dist.ReduceOp.SUM, group=group)
        global_sizes = [int(x) for x in gs.cpu()]
    yield {"global_sizes": global_sizes}

    def issue(i):
        """start the all-to-all-v of file i (asynchronous: it overlaps the merge of file i - 1)"""
        k = files_keys[i]
        rcv = recv_counts_all[i]
        out = torch.empty(sum(rcv), dtype=k.dtype, device=k.device)
        works = [_all_to_all(out, k, rcv, list(counts_all[i]), group, async_op=True)]
        out_t = None
        if files_taxids is not None:
            t = files_taxids[i]
            out_t = torch.empty(sum(rcv), dtype=t.dtype, device=t.device)
            works.append(_all_to_all(out_t, t, rcv, list(counts_all[i]), group, async_op=True))
        return works, out, out_t

    pending = issue(0) if nfiles else None
    for i in range(nfiles):
        nxt = issue(i + 1) if i + 1 < nfiles else None      # file i+1 travels while file i is merged
        works, rk, rt = pending
        for w in works:
            w.wait()   # nccl: makes torch's current stream (= the ctx stream) wait, the host does not block
        pieces = split_by_counts(rk, recv_counts_all[i])
        tpieces = split_by_counts(rt, recv_counts_all[i]) if rt is not None else None
        yield pieces, tpieces
        pending = nxt


def pieces_in_value_order(recv, counts):
    """True if the received buffer `recv` (the slices of all source ranks back to back, `counts` records each) is
    already sorted as a whole: the last code of every non-empty slice is <= the first code of the next one.  That is the
    case whenever the logical file was OFFSET-sharded (rank i held the i-th contiguous chunk of the sorted file: SURVEY
    8(e)'s own example) -- the rebuilt file is then the buffer itself and no merge pass runs.  Costs one 2 * world
    element read-back."""
    idx, off = [], 0
    for c in counts:
        if c:
            idx += [off, off + c - 1]
        off += c
    if len(idx) <= 2:
        return True
    ends = recv[torch.tensor(idx, dtype=torch.int64, device=recv.device)].cpu().tolist()
    # uint64 bit patterns in int64 tensors: compare as unsigned
    u = [x & 0xFFFFFFFFFFFFFFFF for x in ends]
    return all(u[i] <= u[i + 1] for i in range(1, len(u) - 1, 2))


def redistribute_pipelined(ctx, files_keys, key_bits, files_taxids=None, group=None, with_sizes=False, splitters=None, chunks=4):
    """redistribute() with every rank's range cut into `chunks` sub-ranges BY VALUE and the exchange done sub-range by
    sub-range: while the slices of sub-range q that have arrived are rebuilt (k-way merge of one piece per source rank, or
    nothing when they arrive in value order), the all-to-all-v of sub-range q + 1 is already travelling (SURVEY 8(e):
    "overlap per-peer transfers with merging of already-arrived ranges").  With two files (the metric) the plain pipeline
    only hides file B's exchange behind file A's rebuild; here every exchange but the first hides behind a rebuild, and a
    rank's peak receive buffer is 1 / chunks of a file.  The rebuilt sub-ranges are written side by side into one
    preallocated tensor per file (they are disjoint, ascending value ranges), so the result is what redistribute() returns,
    bit for bit.  One small all-to-all carries the slice sizes of all files and sub-ranges."""
    world = dist.get_world_size(group)
    Q = max(1, int(chunks))
    spl_full = _resolve_splitters(files_keys, key_bits, splitters, group)
    fine = []
    for g in range(world):
        lo, hi = int(spl_full[g]), int(spl_full[g + 1])
        fine += [lo + ((hi - lo) * q) // Q for q in range(Q)]
    nfiles = len(files_keys)
    if nfiles == 0:
        return ([], ([] if files_taxids is not None else None), []) if with_sizes else ([], ([] if files_taxids is not None else None))
    dev = files_keys[0].device
    cuts = []
    for k in files_keys:
        c = [int(x) for x in ctx.partition_points(k, fine)] + [k.numel()]
        cuts.append(c)
    # counts[i][g][q]: records of my file i for sub-range q of rank g.  Beside every count travel the slice's FIRST and LAST
    # code (one device gather per file, no host round trip): the receiver decides "the pieces arrive in value order" for
    # every unit from this one all-to-all and its one read-back, instead of a 2 * world element read-back + stream sync per
    # (file, sub-range) unit on the critical path of the pipeline (round-5 advice).
    U3 = nfiles * Q
    cnt = torch.tensor([[cuts[i][g * Q + q + 1] - cuts[i][g * Q + q] for i in range(nfiles) for q in range(Q)] for g in range(world)],
                       dtype=torch.int64, device=dev)                        # [dest rank][file * Q + q]
    firsts = torch.zeros((world, U3), dtype=torch.int64, device=dev)
    lasts = torch.zeros((world, U3), dtype=torch.int64, device=dev)
    for i, k in enumerate(files_keys):
        if k.numel() == 0:
            continue
        c = cuts[i]
        lo = torch.tensor([[min(c[g * Q + q], k.numel() - 1) for q in range(Q)] for g in range(world)], dtype=torch.int64, device=dev)
        hi = torch.tensor([[max(min(c[g * Q + q + 1], k.numel()) - 1, 0) for q in range(Q)] for g in range(world)], dtype=torch.int64, device=dev)
        firsts[:, i * Q:(i + 1) * Q] = k[lo.flatten()].view(world, Q)
        lasts[:, i * Q:(i + 1) * Q] = k[hi.flatten()].view(world, Q)
    send = torch.cat([cnt, firsts, lasts], dim=1).contiguous()               # [dest rank][counts | firsts | lasts]
    recv = torch.empty_like(send)
    _all_to_all(recv, send, group=group)
    rc = recv.cpu().tolist()                                                 # [source rank][...]: the ONE read-back
    global_sizes = None
    if with_sizes:
        gs = torch.tensor([k.numel() for k in files_keys], dtype=torch.int64, device=dev if dist.get_backend(group) != "gloo" else "cpu")
        dist.all_reduce(gs, op=dist.ReduceOp.SUM, group=group)
        global_sizes = [int(x) for x in gs.cpu()]
    units = [(i, q) for i in range(nfiles) for q in range(Q)]

    def counts_of(i, q):
        return [int(rc[src][i * Q + q]) for src in range(world)]

    def in_value_order(i, q):
        """pieces_in_value_order() from the exchanged ends: last code of every non-empty piece <= first code of the next one"""
        prev = None
        for src in range(world):
            if not rc[src][i * Q + q]:
                continue
            first = rc[src][U3 + i * Q + q] & 0xFFFFFFFFFFFFFFFF             # (uint64 bit patterns in int64 tensors)
            if prev is not None and prev > first:
                return False
            prev = rc[src][2 * U3 + i * Q + q] & 0xFFFFFFFFFFFFFFFF
        return True

    def issue(i, q):
        k = files_keys[i]
        c = cuts[i]
        ins = [c[g * Q + q + 1] - c[g * Q + q] for g in range(world)]
        # the sub-range's slices, destination by destination (one sub-range per destination: they already lie back to back)
        sk = k[c[0]:c[world]] if Q == 1 else torch.cat([k[c[g * Q + q]:c[g * Q + q + 1]] for g in range(world)])
        rcv = counts_of(i, q)
        out = torch.empty(sum(rcv), dtype=k.dtype, device=k.device)
        works = [_all_to_all(out, sk, rcv, ins, group, async_op=True)]
        keep = [sk]
        out_t = None
        if files_taxids is not None:
            t = files_taxids[i]
            st = t[c[0]:c[world]] if Q == 1 else torch.cat([t[c[g * Q + q]:c[g * Q + q + 1]] for g in range(world)])
            out_t = torch.empty(sum(rcv), dtype=t.dtype, device=t.device)
            works.append(_all_to_all(out_t, st, rcv, ins, group, async_op=True))
            keep.append(st)
        return works, out, out_t, keep
    totals = [sum(sum(counts_of(i, q)) for q in range(Q)) for i in range(nfiles)]
    local = [torch.empty(totals[i], dtype=files_keys[i].dtype, device=dev) for i in range(nfiles)]
    local_t = [torch.empty(totals[i], dtype=files_taxids[i].dtype, device=dev) for i in range(nfiles)] if files_taxids is not None else None
    offs = [0] * nfiles
    pending = issue(*units[0])
    for u, (i, q) in enumerate(units):
        nxt = issue(*units[u + 1]) if u + 1 < len(units) else None             # sub-range u + 1 travels while u is rebuilt
        works, rk, rt, _keep = pending
        for w in works:
            w.wait()
        counts = counts_of(i, q)
        n = sum(counts)
        dst = local[i][offs[i]:offs[i] + n]
        dst_t = local_t[i][offs[i]:offs[i] + n] if local_t is not None else None
        if n:
            if in_value_order(i, q):
                dst.copy_(rk)
                if dst_t is not None:
                    dst_t.copy_(rt)
            else:
                pieces = split_by_counts(rk, counts)
                tpieces = split_by_counts(rt, counts) if rt is not None else None
                merged = ctx.merge_k(pieces, tpieces, out=dst, out_taxids=dst_t) if dst_t is not None else ctx.merge_k(pieces, None, out=dst)
                mk = merged[0] if dst_t is not None else merged
                if mk.data_ptr() != dst.data_ptr():                             # (a context that ignores `out`)
                    dst.copy_(mk)
                    if dst_t is not None:
                        dst_t.copy_(merged[1])
        offs[i] += n
        pending = nxt
    return (local, local_t, global_sizes) if with_sizes else (local, local_t)


def redistribute(ctx, files_keys, key_bits, files_taxids=None, group=None, with_sizes=False, splitters=None, pipeline=1):
    if pipeline and int(pipeline) > 1:
        return redistribute_pipelined(ctx, files_keys, key_bits, files_taxids, group, with_sizes, splitters, int(pipeline))
    return _redistribute_whole(ctx, files_keys, key_bits, files_taxids, group, with_sizes, splitters)


def _redistribute_whole(ctx, files_keys, key_bits, files_taxids=None, group=None, with_sizes=False, splitters=None):
    """Prefix redistribution of FILE-sharded sorted files: every logical file is exchanged ONCE and rebuilt on its
    range owner.  Slices that arrive already in value order (offset-sharded inputs) are the rebuilt file as they lie in
    the receive buffer; otherwise the slices, which overlap in value, go through the keep-EVERYTHING k-way merge
    (mergeChunksFile's order: equal codes by source rank) -- not a union, so a file that holds a code twice still does
    afterwards and `inter` / `diff` keep the reference's multiset behaviour (inter.go:228-257) across ranks.  Returns the
    list of local sorted files (and the list of their taxids, or None): the inputs of any number of single-GPU
    operations on this rank's range -- `union` AND `inter` of the same two sets need one exchange of each, not one per
    operation.

    PRECONDITION: the per-rank chunks of one logical file are DISJOINT (each record of the file lives on exactly one rank:
    stride-, offset- or hash-sharded reads of one .unik file).  Chunks that overlap would come back as duplicates inside
    the rebuilt file -- the keep-everything merge cannot tell them from a file that really holds a code twice -- and
    `common` would then count such a code once per chunk.  Callers whose chunks may overlap rebuild with ctx.union instead.
    (The value-order check costs one 2 * world element read-back per file on the exchange pipeline.)"""
    it = _exchange_files(ctx, files_keys, key_bits, files_taxids, group, splitters)
    info = next(it)
    local, local_t = [], ([] if files_taxids is not None else None)
    for pieces, tpieces in it:
        counts = [x.numel() for x in pieces]
        # (the pieces are views of one receive buffer, back to back in source-rank order)
        whole = pieces[0] if len(pieces) == 1 else _joined(pieces)
        if whole is not None and pieces_in_value_order(whole, counts):
            local.append(whole)
            if tpieces is not None:
                local_t.append(tpieces[0] if len(tpieces) == 1 else _joined(tpieces))
            continue
        merged = ctx.merge_k(pieces, tpieces)              # PLAIN: every record kept
        if tpieces is not None:
            local.append(merged[0])
            local_t.append(merged[1])
        else:
            local.append(merged)
    return (local, local_t, info["global_sizes"]) if with_sizes else (local, local_t)


def comm_init_from_dist(ctx, group=None):
    """Bring up the LIBRARY'S OWN RCCL communicator (ukm_comm_init, what a Go / C host calls) for the ranks of a
    torch.distributed group: rank 0 makes the unique id, torch.distributed hands it to the others (the reference host would
    use a file or a socket), every rank calls ukm_comm_init -- collective."""
    from . import lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [lib.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init(world, rank, box[0])


def redistribute_cabi(ctx, files_keys, key_bits, files_taxids=None, splitters=None):
    """redistribute() through the C ABI alone -- ukm_partition_points, ONE ukm_shard_counts for all files, one
    ukm_shard_exchange_known per file (grouped ncclSend / ncclRecv inside the library, ukm_comm.hip), the rebuild by
    ukm_merge_k -- i.e. exactly the calls of INTEGRATION.md's Go loop; torch.distributed plays no part (the context's
    communicator must be up: comm_init_from_dist).  splitters: None = equal-width prefix ranges, or world + 1 boundaries."""
    world, _ = ctx.comm_info()
    spl = (prefix_splitters(key_bits, world) if splitters is None else list(splitters))[:-1]
    nfiles = len(files_keys)
    if nfiles == 0:
        return [], ([] if files_taxids is not None else None)
    import numpy as np
    send = np.array([cuts_to_counts(ctx.partition_points(k, spl), k.numel()) for k in files_keys], dtype=np.uint64)
    # [file][source rank]: one gather, one host round trip; the "comes with taxids" flags ride along (ranks that disagree raise)
    recv = ctx.shard_counts(send, [files_taxids is not None and files_taxids[i] is not None for i in range(nfiles)])
    local, local_t = [], ([] if files_taxids is not None else None)
    for i, k in enumerate(files_keys):
        t = files_taxids[i] if files_taxids is not None else None
        rk, rt, rc = ctx.shard_exchange(k, send[i], t, recv_counts=recv[i])
        counts = [int(x) for x in rc]
        if pieces_in_value_order(rk, counts):
            local.append(rk)
            if rt is not None:
                local_t.append(rt)
            continue
        pieces = split_by_counts(rk, counts)
        tpieces = split_by_counts(rt, counts) if rt is not None else None
        merged = ctx.merge_k(pieces, tpieces)
        if tpieces is not None:
            local.append(merged[0])
            local_t.append(merged[1])
        else:
            local.append(merged)
    return local, local_t


def _joined(pieces):
    """the tensor the back-to-back views `pieces` were cut from (split_by_counts), or None if they are not adjacent"""
    base = pieces[0]
    total = sum(x.numel() for x in pieces)
    try:
        whole = base.as_strided((total,), (1,))
    except RuntimeError:
        return None
    off = 0
    for x in pieces:
        if x.numel() and x.data_ptr() != whole.data_ptr() + off * whole.element_size():
            return None
        off += x.numel()
    return whole


def sharded_setop(ctx, op, files_keys, key_bits, files_taxids=None, group=None, splitters=None, **kw):
    """`union` / `inter` / `diff` / `common` over files that are FILE-sharded across ranks
    (every rank holds whole sorted files spanning the full code range).

    1. cut every local file at the prefix splitters (GPU lower_bound),
    2. one all-to-all-v per file ships slice g to rank g,
    3. `union`: all received pieces feed one k-way union.  `inter` / `diff` / `common` fold over FILES, so each rank
       first rebuilds its range of every logical file (k-way merge of the pieces that arrived for it: `redistribute`),
    4. then the single-GPU n-way op runs on the rank's range.
    Returns this rank's part of the result; concatenating the parts in rank order gives the
    globally sorted output, bit-identical to the 1-GPU result (including `inter`'s empty-later-file rule, which
    is decided on the global file sizes, not on a rank's slice).

    splitters: None = equal-width prefix ranges (right for hashes), "sampled" = boundaries from a sample of all ranks'
    files that balance the ranks' record counts (k-mer codes), or an explicit list of world + 1 boundaries.

    The per-rank chunks of one logical file must be disjoint (see redistribute).

    files_keys: list of 1-D int64 device tensors; every rank must pass the SAME number of
    files (file i of every rank are the per-rank chunks of logical input i).  `ctx` must run on
    torch's current stream (lib.Context(device, stream=torch.cuda.current_stream().cuda_stream)):
    the exchange of file i+1 is issued asynchronously and overlaps the merge of file i.
    """
    nfiles = len(files_keys)
    if nfiles == 0:
        raise ValueError("sharded_setop: no input files (every rank passes the same, non-zero number of files)")
    if op == "union":
        # a union does not care which logical file a record came from: every received piece (nfiles x world sorted
        # slices of this rank's range) goes straight into ONE k-way union -- no per-file merge pass over HBM first
        it = _exchange_files(ctx, files_keys, key_bits, files_taxids, group, splitters)
        next(it)
        allp, allt = [], ([] if files_taxids is not None else None)
        for pieces, tpieces in it:
            for j, x in enumerate(pieces):
                if x.numel():
                    allp.append(x)
                    if tpieces is not None:
                        allt.append(tpieces[j])
        if not allp:
            empty = files_keys[0][:0]
            return (empty, files_taxids[0][:0]) if files_taxids is not None else empty
        return ctx.union(allp, allt)
    local, local_t, global_sizes = redistribute(ctx, files_keys, key_bits, files_taxids, group, with_sizes=True,
                                                splitters=splitters)
    fn = {"inter": ctx.inter, "diff": ctx.diff, "common": ctx.common}[op]
    if op == "common":
        return fn(local, kw["threshold"], local_t)
    if op == "inter" and nfiles > 1:
        # The reference's `inter` stops at an EMPTY later file and keeps the running result (inter.go:211-217;
        # ukm_inter reproduces it).  That decision is about the GLOBAL file: a rank whose slice of file i is
        # empty while the file is not must return nothing for its range, and every rank must stop at the
        # first globally empty later file.
        stop = next((i for i in range(1, nfiles) if global_sizes[i] == 0), nfiles)
        local = local[:stop]
        if local_t is not None:
            local_t = local_t[:stop]
        if any(local[i].numel() == 0 for i in range(1, stop)):
            empty = local[0][:0]
            return (empty, local_t[0][:0]) if local_t is not None else empty
    return fn(local, local_t)


def sharded_sort(ctx, keys, key_bits, taxids=None, group=None, splitters=None):
    """Distributed `sort` of UNSORTED codes that are spread over the ranks in any way (the count path:
    every rank encoded its own reads).  Each rank sorts what it holds, cuts the sorted stream at the
    prefix splitters, one all-to-all-v moves slice g to rank g, and the received sorted slices (one per
    source rank, all inside this rank's value range) are combined by the k-way merge.  Returns this
    rank's range, sorted; the concatenation over ranks in rank order is the global sort.  Records with
    equal codes keep (source rank, local position) order, i.e. the sort is stable w.r.t. rank order.

    keys: 1-D int64 device tensor (uint64 bit patterns), modified in place by the local sort.
    """
    world = dist.get_world_size(group)
    if taxids is not None:
        ctx.sort_pairs(keys, taxids, key_bits)
    else:
        ctx.sort_u64(keys, key_bits)
    spl = _resolve_splitters([keys], key_bits, splitters, group)[:-1]   # ("sampled": of the locally sorted codes)
    cuts = ctx.partition_points(keys, spl)
    counts = cuts_to_counts(cuts, keys.numel())
    rk, rt, rc = exchange_sorted(keys, counts, taxids, group)
    pieces = split_by_counts(rk, rc)
    tpieces = split_by_counts(rt, rc) if rt is not None else None
    return ctx.merge_k(pieces, tpieces)          # PLAIN: every record kept


def sharded_count(ctx, keys, key_bits, mode=1, taxids=None, group=None, splitters=None):
    """`count` after the per-rank encode: distinct codes (mode 1 = UNIQUE; 2 = repeated, 4 = singleton as
    in include/unikmer_hip.h) of everything all ranks hold, range-partitioned by prefix."""
    merged = sharded_sort(ctx, keys, key_bits, taxids, group, splitters)
    if taxids is not None:
        return ctx.unique(merged[0], merged[1], mode=mode)
    return ctx.unique(merged, mode=mode)
