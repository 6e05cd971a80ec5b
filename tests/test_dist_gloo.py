"""World-size-2 CPU tests (gloo) of the multi-GPU prefix-sharding plumbing
(unikmer_amd/dist.py).  The GPU library is not involved: cut points are computed with numpy
HERE (on the GPU they come from ukm_partition_points) so that only the all-to-all-v exchange,
the splitter arithmetic and the "concatenate in rank order" contract are under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unikmer_amd import dist as ud


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _files(rank, nfiles=3, n=20000, key_bits=42):
    """every rank holds `nfiles` whole sorted files spanning the full code range"""
    out = []
    for f in range(nfiles):
        rng = np.random.default_rng(1000 * f + rank)
        out.append(np.unique(rng.integers(0, 1 << key_bits, n, dtype=np.uint64)))
    return out


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        key_bits = 42
        spl = ud.prefix_splitters(key_bits, world)
        assert spl[0] == 0 and spl[-1] == 1 << key_bits and len(spl) == world + 1
        files = _files(rank)
        mine = []
        for k in files:
            cuts = np.searchsorted(k, np.array(spl[:-1], dtype=np.uint64), side="left")  # = ukm_partition_points
            counts = ud.cuts_to_counts(cuts, len(k))
            t = torch.from_numpy(k.view(np.int64))
            tax = torch.arange(len(k), dtype=torch.int32) + 1000000 * rank
            rk, rt, rc = ud.exchange_sorted(t, counts, tax)
            pieces = ud.split_by_counts(rk, rc)
            tpieces = ud.split_by_counts(rt, rc)
            assert len(pieces) == world
            for src, (p, tp) in enumerate(zip(pieces, tpieces)):
                v = p.numpy().view(np.uint64)
                assert np.all(v[1:] > v[:-1])                       # each slice stays sorted
                assert np.all(v >= np.uint64(spl[rank])) and np.all(v < np.uint64(spl[rank + 1]))
                assert np.all(tp.numpy() // 1000000 == src)          # payload travelled with its keys
            mine.append(np.unique(np.concatenate([p.numpy().view(np.uint64) for p in pieces])))
        # this rank's part of the global union / intersection of the 3 logical files
        u = mine[0]
        i = mine[0]
        for m in mine[1:]:
            u = np.union1d(u, m)
            i = np.intersect1d(i, m)
        ret[rank] = (u, i)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_prefix_exchange_world2(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    # reference: the same logical files (file f = union over ranks of rank-local chunk f)
    logical = []
    for f in range(3):
        logical.append(np.unique(np.concatenate([_files(r)[f] for r in range(world)])))
    gu = logical[0]
    gi = logical[0]
    for m in logical[1:]:
        gu = np.union1d(gu, m)
        gi = np.intersect1d(gi, m)
    # concatenation of the ranks' parts in rank order IS the global sorted result
    assert np.array_equal(np.concatenate([ret[r][0] for r in range(world)]), gu)
    assert np.array_equal(np.concatenate([ret[r][1] for r in range(world)]), gi)


def test_splitters_and_counts():
    s = ud.prefix_splitters(62, 8)
    assert s[1] - s[0] == (1 << 62) // 8 and s[-1] == 1 << 62
    s64 = ud.prefix_splitters(64, 4)
    assert s64[-1] == (1 << 64) - 1 and s64[1] == 1 << 62
    assert ud.cuts_to_counts([0, 3, 3, 10], 12) == [3, 0, 7, 2]


class _NumpyCtx:
    """Stand-in for unikmer_amd.lib.Context in the CPU tests of the distributed plumbing: the four calls
    dist.sharded_sort / sharded_count make, on torch CPU tensors (test infrastructure only; on the GPU the
    same calls go to libunikmer_hip.so)."""

    @staticmethod
    def _u(t):
        return t.numpy().view(np.uint64)

    def sort_u64(self, keys, key_bits=64):
        keys.copy_(torch.from_numpy(np.sort(self._u(keys)).view(np.int64)))
        return keys

    def sort_pairs(self, keys, vals, key_bits=64):
        o = np.argsort(self._u(keys), kind="stable")
        k2, v2 = self._u(keys)[o].copy(), vals.numpy()[o].copy()
        keys.copy_(torch.from_numpy(k2.view(np.int64)))
        vals.copy_(torch.from_numpy(v2))
        return keys, vals

    def partition_points(self, keys, splitters):
        return np.searchsorted(self._u(keys), np.array(splitters, dtype=np.uint64), side="left")

    def merge_k(self, pieces, tpieces=None):
        cat = np.concatenate([self._u(p) for p in pieces]) if pieces else np.empty(0, np.uint64)
        o = np.argsort(cat, kind="stable")
        k = torch.from_numpy(cat[o].view(np.int64))
        if tpieces is None:
            return k
        t = np.concatenate([p.numpy() for p in tpieces])[o]
        return k, torch.from_numpy(t)

    def unique(self, keys, taxids=None, mode=1):
        assert taxids is None and mode == 1
        return torch.from_numpy(np.unique(self._u(keys)).view(np.int64))

    # n-way set operations on sorted sets (no taxids in these plumbing tests)
    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64))

    def union(self, files, taxids=None):
        out = np.empty(0, np.uint64)
        for f in files:
            out = np.union1d(out, self._u(f))
        return self._t(out)

    def inter(self, files, taxids=None):
        # ukm_inter's semantics incl. the reference quirk: an EMPTY later stream ends the fold and the running
        # result is kept (inter.go:211-217); an empty first stream gives nothing
        out = self._u(files[0])
        for f in files[1:]:
            if len(out) == 0 or f.numel() == 0:
                break
            out = np.intersect1d(out, self._u(f))
        return self._t(out)

    def diff(self, files, taxids=None):
        out = self._u(files[0])
        for f in files[1:]:
            out = np.setdiff1d(out, self._u(f))
        return self._t(out)

    def common(self, files, threshold, taxids=None):
        cat = np.concatenate([np.unique(self._u(f)) for f in files])
        v, c = np.unique(cat, return_counts=True)
        return self._t(v[c >= threshold])


def _sort_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(77 + rank)
        x = rng.integers(0, 1 << 42, 30000 + 1000 * rank, dtype=np.uint64)
        x[:500] = x[500:1000]                                  # duplicates, also across ranks below
        if rank == 1:
            x[1000:1500] = np.random.default_rng(77).integers(0, 1 << 42, 30000, dtype=np.uint64)[1000:1500]
        keys = torch.from_numpy(x.copy().view(np.int64))
        tax = torch.arange(len(x), dtype=torch.int32) + 1000000 * rank
        ctx = _NumpyCtx()
        sk, st = ud.sharded_sort(ctx, keys.clone(), 42, tax.clone())
        su = ud.sharded_count(ctx, keys.clone(), 42)
        ret[rank] = (x, sk.numpy().view(np.uint64).copy(), st.numpy().copy(), su.numpy().view(np.uint64).copy())
    finally:
        dist.destroy_process_group()


def test_sharded_sort_and_count_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sort_worker, args=(world, port, ret), nprocs=world, join=True)
    allx = np.concatenate([ret[r][0] for r in range(world)])
    gk = np.concatenate([ret[r][1] for r in range(world)])
    gt = np.concatenate([ret[r][2] for r in range(world)])
    assert np.array_equal(gk, np.sort(allx))                  # concatenation in rank order = global sort
    # every payload still sits next to its key: taxid encodes (rank, position)
    src = gt // 1000000
    pos = gt % 1000000
    assert all(np.array_equal(ret[r][0][pos[src == r]], gk[src == r]) for r in range(world))
    # equal codes keep rank order (stable w.r.t. ranks)
    same = gk[1:] == gk[:-1]
    assert np.all(src[1:][same] >= src[:-1][same])
    assert np.array_equal(np.concatenate([ret[r][3] for r in range(world)]), np.unique(allx))
    spl = ud.prefix_splitters(42, world)
    for r in range(world):
        v = ret[r][1]
        assert np.all(v >= np.uint64(spl[r])) and (r == world - 1 or np.all(v < np.uint64(spl[r + 1])))


def _setop_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        files = [torch.from_numpy(f.view(np.int64)) for f in _files(rank)]
        ctx = _NumpyCtx()
        out = {}
        for op in ("union", "inter", "diff"):
            out[op] = ud.sharded_setop(ctx, op, files, 42).numpy().view(np.uint64).copy()
        out["common"] = ud.sharded_setop(ctx, "common", files, 42, threshold=2).numpy().view(np.uint64).copy()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_sharded_setop_pipelined_world2():
    """dist.sharded_setop end to end over gloo: batched count exchange, asynchronous per-file all-to-all-v
    overlapped with the merges, per-rank n-way op; concatenation in rank order == the global result."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_setop_worker, args=(world, port, ret), nprocs=world, join=True)
    logical = [np.unique(np.concatenate([_files(r)[f] for r in range(world)])) for f in range(3)]
    gu, gi, gd = logical[0], logical[0], logical[0]
    for m in logical[1:]:
        gu, gi, gd = np.union1d(gu, m), np.intersect1d(gi, m), np.setdiff1d(gd, m)
    v, c = np.unique(np.concatenate(logical), return_counts=True)
    cat = lambda op: np.concatenate([ret[r][op] for r in range(world)])
    assert np.array_equal(cat("union"), gu) and np.array_equal(cat("inter"), gi) and np.array_equal(cat("diff"), gd)
    assert np.array_equal(cat("common"), v[c >= 2])


def _inter_quirk_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = _NumpyCtx()
        hi = np.uint64((1 << 41) + 1)       # belongs to rank 1's prefix range of a 42-bit code space
        mk = lambda *v: torch.from_numpy(np.array(v, dtype=np.uint64).view(np.int64))
        out = {}
        # case 1: A = {1, 2^41+1}, B = {1}: rank 1's slice of B is empty, B is not -> rank 1 returns nothing
        A = mk(1, hi) if rank == 0 else mk()
        B = mk(1) if rank == 0 else mk()
        out["slice_empty"] = ud.sharded_setop(ctx, "inter", [A, B], 42).numpy().view(np.uint64).copy()
        # case 2: B is GLOBALLY empty -> the reference stops and keeps A (every rank keeps its range of A), C is never looked at
        C3 = mk(7) if rank == 1 else mk()
        out["file_empty"] = ud.sharded_setop(ctx, "inter", [A, mk(), C3], 42).numpy().view(np.uint64).copy()
        # case 3: the chunks of B live on the OTHER rank
        A2 = mk(1, 5, hi) if rank == 0 else mk(9)
        B2 = mk() if rank == 0 else mk(5, 9, hi)
        out["cross"] = ud.sharded_setop(ctx, "inter", [A2, B2], 42).numpy().view(np.uint64).copy()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_sharded_inter_empty_slice_vs_empty_file_world2():
    """`inter` stops at an empty later FILE (inter.go:211-217), not at an empty per-rank slice of a file."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_inter_quirk_worker, args=(world, port, ret), nprocs=world, join=True)
    cat = lambda key: np.concatenate([ret[r][key] for r in range(world)])
    hi = (1 << 41) + 1
    assert cat("slice_empty").tolist() == [1]
    assert cat("file_empty").tolist() == [1, hi]
    assert cat("cross").tolist() == [5, 9, hi]
