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
