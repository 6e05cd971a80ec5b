"""The multi-GPU path (unikmer_amd/dist.py) with the REAL HIP context under it and a world of two ranks.
A gpurun box has one GPU, so both ranks share cuda:0 and the collectives run over gloo (dist._all_to_all
stages device tensors through the host for that backend only); everything else — ukm_partition_points, the
per-rank k-way merges and n-way set operations — is the product path on the device.  The same code runs
over `nccl` (= RCCL) with one rank per GPU; test_gpu_parity.py has the world-1 nccl run."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY_BITS = 42


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _files(rank, nfiles=4, n=150_000):
    out = []
    for f in range(nfiles):
        rng = np.random.default_rng(1000 * f + rank)
        # the two ranks' chunks of one logical file overlap in value and share some codes
        base = np.random.default_rng(500 + f).integers(0, 1 << KEY_BITS, n // 3, dtype=np.uint64)
        out.append(np.unique(np.concatenate([rng.integers(0, 1 << KEY_BITS, n, dtype=np.uint64), base])))
    return out


def _taxids(keys, salt, T):
    """a taxid per CODE (the same on both ranks), different per file"""
    with np.errstate(over="ignore"):
        h = (keys + np.uint64(salt)) * np.uint64(0x9E3779B97F4A7C15)
    return ((h >> np.uint64(40)) % np.uint64(T) + np.uint64(1)).astype(np.uint32)


def _worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import synth_tree
    from unikmer_amd import dist as ud
    from unikmer_amd import lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        child, parent = synth_tree(4, 8)
        ctx.taxonomy_load(child, parent)
        T = len(child)
        up = lambda a, dt=np.int64: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
        down = lambda t: t.cpu().numpy().view(np.uint64).copy()
        files = _files(rank)
        dfiles = [up(f) for f in files]
        dtax = [up(_taxids(f, i, T), np.int32) for i, f in enumerate(files)]
        out = {}
        for op in ("union", "inter", "diff"):
            out[op] = down(ud.sharded_setop(ctx, op, dfiles, KEY_BITS))
            k, t = ud.sharded_setop(ctx, op, dfiles, KEY_BITS, files_taxids=dtax)
            out[op + "_t"] = (down(k), t.cpu().numpy().view(np.uint32).copy())
        out["common"] = down(ud.sharded_setop(ctx, "common", dfiles, KEY_BITS, threshold=3))
        # inter: empty per-rank slice vs globally empty later file (inter.go:211-217)
        hi = (1 << (KEY_BITS - 1)) + 1
        mk = lambda *v: up(np.array(v, dtype=np.uint64))
        A = mk(1, hi) if rank == 0 else mk()
        B = mk(1) if rank == 0 else mk()
        out["slice_empty"] = down(ud.sharded_setop(ctx, "inter", [A, B], KEY_BITS))
        out["file_empty"] = down(ud.sharded_setop(ctx, "inter", [A, mk(), mk(7) if rank else mk()], KEY_BITS))
        # the count path: unsorted codes on both ranks -> distributed sort / distinct set
        rng = np.random.default_rng(77 + rank)
        x = rng.integers(0, 1 << KEY_BITS, 200_000 + 999 * rank, dtype=np.uint64)
        x[:2000] = np.random.default_rng(5).integers(0, 1 << KEY_BITS, 2000, dtype=np.uint64)   # shared across ranks
        out["x"] = x
        out["sorted"] = down(ud.sharded_sort(ctx, up(x.copy()), KEY_BITS))
        out["distinct"] = down(ud.sharded_count(ctx, up(x.copy()), KEY_BITS))
        ret[rank] = out
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sharded_ops_two_ranks_real_context():
    import torch.multiprocessing as mp
    from conftest import synth_tree
    from oracle import oracle as O
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    child, parent = synth_tree(4, 8)
    tax = O.Taxonomy(child, parent)
    T = len(child)
    nfiles = 4
    # logical file f = union over ranks of the rank-local chunks (LCA-folded where both ranks hold a code)
    logical, logical_t = [], []
    for f in range(nfiles):
        chunks = [_files(r)[f] for r in range(world)]
        k, t = O.union(chunks, [_taxids(c, f, T) for c in chunks], tax)
        logical.append(k)
        logical_t.append(t)
    cat = lambda key: np.concatenate([ret[r][key] for r in range(world)])
    catt = lambda key, j: np.concatenate([ret[r][key][j] for r in range(world)])
    for op, ofn in (("union", O.union), ("inter", O.inter), ("diff", O.diff)):
        assert np.array_equal(cat(op), ofn(logical)), op
        ek, et = ofn(logical, logical_t, tax)
        assert np.array_equal(catt(op + "_t", 0), ek) and np.array_equal(catt(op + "_t", 1), et), op
    assert np.array_equal(cat("common"), O.common(logical, 3))
    hi = (1 << (KEY_BITS - 1)) + 1
    assert cat("slice_empty").tolist() == [1]
    assert cat("file_empty").tolist() == [1, hi]
    allx = np.concatenate([ret[r]["x"] for r in range(world)])
    assert np.array_equal(cat("sorted"), np.sort(allx))
    assert np.array_equal(cat("distinct"), np.unique(allx))


def _kmer_worker(rank, world, port, path, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from unikmer_amd import dist as ud
    from unikmer_amd import lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        codes = np.load(path)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
        down = lambda t: t.cpu().numpy().view(np.uint64).copy()
        A = codes[rank::world]                                   # stride-sharded
        Bf = codes[::3]
        B = Bf[rank * len(Bf) // world:(rank + 1) * len(Bf) // world]   # offset-sharded: arrives in value order
        M = np.sort(np.concatenate([codes[::7], codes[::35]]))   # a file that holds every 5th of its codes twice
        files = [up(A), up(B), up(M[rank::world])]
        out = {"spl": ud.sampled_splitters(files, 62)}
        for name, sp in (("equal", None), ("sampled", "sampled")):
            local, _ = ud.redistribute(ctx, files, 62, splitters=sp)
            out[name] = [down(x) for x in local]
            out[name + "_inter"] = down(ud.sharded_setop(ctx, "inter", files, 62, splitters=sp))
            out[name + "_diff"] = down(ud.sharded_setop(ctx, "diff", [files[0], files[2]], 62, splitters=sp))
        ret[rank] = out
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_sampled_splitters_two_ranks_real_context(tmp_path):
    """SURVEY 8(e) on real k-mer codes (distinct canonical 31-mers of the E. coli fixture) through the HIP context:
    equal-width ranges load the two ranks unevenly, sampled splitters bring both within 10 % of the mean, the boundaries
    agree across ranks, and every result -- rebuilt files (a multiset file keeps its duplicates), inter with the
    reference's multiset rule, diff -- concatenates to the 1-GPU answer of the oracle."""
    import torch.multiprocessing as mp
    from conftest import read_fasta_gz, MG1655
    from oracle import oracle as O
    seq, off = read_fasta_gz(MG1655)
    codes = np.unique(O.count_windows(seq, off, 31))
    path = str(tmp_path / "codes.npy")
    np.save(path, codes)
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_kmer_worker, args=(world, port, path, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["spl"] == r1["spl"]
    B = codes[::3]
    M = np.sort(np.concatenate([codes[::7], codes[::35]]))
    want_inter = O.inter([codes, B, M])
    want_diff = O.diff([codes, M])
    for name in ("equal", "sampled"):
        for j, want in enumerate((codes, B, M)):
            assert np.array_equal(np.concatenate([r0[name][j], r1[name][j]]), want), (name, j)
        assert np.array_equal(np.concatenate([r0[name + "_inter"], r1[name + "_inter"]]), want_inter), name
        assert np.array_equal(np.concatenate([r0[name + "_diff"], r1[name + "_diff"]]), want_diff), name
    load = lambda name: np.array([sum(len(x) for x in r0[name]), sum(len(x) for x in r1[name])], dtype=np.float64)
    assert load("equal").max() / load("equal").mean() > 1.2
    assert load("sampled").max() / load("sampled").mean() < 1.10


def _run_bench_two_ranks(extra, world=2):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UKM_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--set-size", "1e6", "--cpu-sample", "0"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_incl_exchange():
    """bench.py's N > 1 path end to end (launched exactly as the driver launches it), two ranks sharing the one
    GPU over gloo.  `value` is the END-TO-END figure (file-sharded start -> each input exchanged once ->
    dist.redistribute -> union + inter), `value_prepartitioned` the same job without the exchange; both legs and the
    bare all-to-all-v are produced and agree on the result sizes (bench.py asserts that)."""
    res = _run_bench_two_ranks([])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert "error" not in res.get("exchange", {}), res.get("exchange")
    assert res["value"] == res["value_incl_exchange"] == res["incl_exchange"]["value"]
    assert res["value_is"].startswith("end to end")
    assert res["value_prepartitioned"] > 0 and res["ms_per_step"] == res["incl_exchange"]["ms_per_step"]
    # the same job from an offset-sharded start (slices arrive in value order: no merge pass), beside the stride one
    off = res["incl_exchange"]["offset_sharded_start"]
    assert "error" not in off, off
    assert off["value"] > 0 and res["value_offset_sharded_start"] == off["value"]
    # the leg through the library's own exchange needs one device per rank (RCCL): recorded as skipped on this box;
    # tests/test_gpu_dist.py::test_redistribute_through_the_c_abi_one_rank runs the same calls on a one-rank communicator
    assert "skipped" in res["incl_exchange"]["exchange_cabi"]
    # the sub-range pipeline (transfers overlap rebuilds) gives the same result sizes and a figure of its own
    pipe = res["incl_exchange"]["pipelined_4_subranges"]
    assert "error" not in pipe, pipe
    assert pipe["value"] > 0 and res["value_pipelined_exchange"] == pipe["value"]
    assert res["incl_exchange"]["steps"] == 2 and res["config"]["per_gpu_set_size"] == 1_000_000
    assert res["config"]["global_set_size"] == 2_000_000
    assert res["roofline"]["frac"] > 0 and res["cpu_baseline"] is None
    assert res["roofline"]["traffic"] is None or res["roofline"]["traffic_from_profile"]["note"].startswith("profile-derived")


def test_bench_eight_ranks_replay_of_the_drivers_launch_line():
    """The driver's N = 8 launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8`)
    replayed on ONE GPU (UKM_BENCH_ONE_GPU: eight ranks share cuda:0, gloo control plane): no 8-GPU node has ever run it
    (SCALE_rNN.json: skipped), so this is what keeps `bench.py --gpus 8` launchable -- the W = 8 splitters, eight-way
    all-to-all-v, per-rank rebuilds and the line's fields."""
    res = _run_bench_two_ranks([], world=8)
    assert res["n_gpus"] == 8 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["per_gpu_set_size"] == 1_000_000 and res["config"]["global_set_size"] == 8_000_000
    assert res["value_is"].startswith("end to end")
    assert res["value"] == res["value_incl_exchange"] == res["incl_exchange"]["value"]
    assert "error" not in res.get("exchange", {}), res.get("exchange")
    assert "skipped" in res["incl_exchange"]["exchange_cabi"]      # (RCCL needs a device per rank)
    assert res["value_prepartitioned"] > 0
    pipe = res["incl_exchange"]["pipelined_4_subranges"]
    assert "error" not in pipe and pipe["value"] > 0, pipe


def test_bench_two_ranks_strong_scaling_and_no_exchange():
    """--scaling strong: --set-size is the GLOBAL set size (the metric's wording), each rank holds 1/N of it;
    --no-exchange: the line says that `value` is the pre-partitioned figure."""
    res = _run_bench_two_ranks(["--scaling", "strong"])
    assert res["scaling"] == "strong" and res["config"]["per_gpu_set_size"] == 500_000
    assert res["config"]["global_set_size"] == 1_000_000 and res["value"] == res["value_incl_exchange"]
    res = _run_bench_two_ranks(["--no-exchange"])
    assert "value_incl_exchange" not in res and res["value"] == res["value_prepartitioned"]
    assert res["value_is"].startswith("PRE-PARTITIONED")


def test_c_abi_rccl_exchange_one_rank():
    """The exchange step behind the C ABI (ukm_comm_* / ukm_shard_exchange: RCCL loaded by the library itself, no
    torch.distributed): a communicator of one rank — the only size a one-GPU box allows (RCCL refuses two ranks on
    one device) — moves host arrays and device tensors through grouped ncclSend / ncclRecv and back."""
    import torch
    from unikmer_amd import dist as ud
    from unikmer_amd import lib
    ctx = lib.Context(0)
    assert ctx.comm_info() == (0, 0)
    uid = lib.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(1, 0, uid)
    assert ctx.comm_info() == (1, 0)
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(0, 1 << 62, 300_000, dtype=np.uint64))
    tax = rng.integers(1, 1000, len(keys)).astype(np.uint32)
    out, out_t, rc = ctx.shard_exchange(keys, [len(keys)], tax)
    assert rc.tolist() == [len(keys)] and np.array_equal(out, keys) and np.array_equal(out_t, tax)
    out, out_t, rc = ctx.shard_exchange(keys, [len(keys)])
    assert out_t is None and np.array_equal(out, keys)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    torch.cuda.synchronize()
    dout, _, rc = ctx.shard_exchange(dk, [len(keys)])
    assert dout.is_cuda and np.array_equal(dout.cpu().numpy().view(np.uint64), keys)
    out, _, rc = ctx.shard_exchange(np.empty(0, np.uint64), [0])
    assert len(out) == 0 and rc.tolist() == [0]
    # the two-step form: sizes of several files in one gather, then exchanges that neither gather nor synchronise
    rcs = ctx.shard_counts([[len(keys)], [7], [0]])
    assert rcs.tolist() == [[len(keys)], [7], [0]]
    out, out_t, rc = ctx.shard_exchange(keys, [len(keys)], tax, recv_counts=rcs[0])
    assert np.array_equal(out, keys) and np.array_equal(out_t, tax)
    # the one-call form (gather of sizes + capacities, collective decision): too small a buffer is an error on all ranks
    L = lib.load()
    import ctypes as C
    small = np.empty(10, dtype=np.uint64)
    rcv = np.zeros(1, dtype=np.uint64)
    sc = np.array([len(keys)], dtype=np.uint64)
    m = C.c_uint64()
    assert L.ukm_shard_exchange(ctx.h, keys.ctypes.data, None, sc.ctypes.data, small.ctypes.data, None, 10, rcv.ctypes.data,
                                C.byref(m)) == lib.ERR_CAPACITY
    assert m.value == len(keys)
    full = np.empty(len(keys), dtype=np.uint64)
    assert L.ukm_shard_exchange(ctx.h, keys.ctypes.data, None, sc.ctypes.data, full.ctypes.data, None, len(full),
                                rcv.ctypes.data, C.byref(m)) == 0
    assert m.value == len(keys) and np.array_equal(full, keys)
    # known counts with a short buffer: the rank still takes part (drains into workspace) and reports the error after
    assert L.ukm_shard_exchange_known(ctx.h, keys.ctypes.data, None, sc.ctypes.data, sc.ctypes.data, small.ctypes.data, None,
                                      10, C.byref(m)) == lib.ERR_CAPACITY
    # ... and so does a rank whose own slice sizes disagree (counts that did not come from ukm_shard_counts)
    wrong = np.array([len(keys) - 1], dtype=np.uint64)
    assert L.ukm_shard_exchange_known(ctx.h, keys.ctypes.data, None, sc.ctypes.data, wrong.ctypes.data, full.ctypes.data, None,
                                      len(full), C.byref(m)) == lib.ERR_INVALID
    assert L.ukm_shard_exchange_known(ctx.h, keys.ctypes.data, None, sc.ctypes.data, sc.ctypes.data, full.ctypes.data, None,
                                      len(full), C.byref(m)) == 0 and np.array_equal(full, keys)
    # sampled splitters (collective; one rank: the boundaries are [0, top]) through the device sampling kernel + all-gather,
    # host arrays and device tensors, an empty file among them
    sp = ctx.shard_splitters([keys, np.empty(0, np.uint64), keys[::2].copy()], 62)
    assert sp == [0, 1 << 62]
    assert ctx.shard_splitters([dk], 64) == [0, (1 << 64) - 1]
    assert ctx.shard_splitters([], 42) == [0, 1 << 42]
    # the library's splitters are dist.py's
    for bits, world in ((62, 8), (42, 3), (64, 4), (2, 5)):
        assert ctx.prefix_splitters(bits, world).tolist() == ud.prefix_splitters(bits, world)[:-1]
    with pytest.raises(lib.UkmError):
        ctx.comm_init(1, 0, uid)            # one communicator per context
    ctx.comm_destroy()
    with pytest.raises(lib.UkmError):
        ctx.shard_exchange(keys, [len(keys)])
    ctx.close()


def test_redistribute_through_the_c_abi_one_rank():
    """dist.redistribute_cabi = INTEGRATION.md's Go loop (ukm_partition_points -> ONE ukm_shard_counts for all files ->
    ukm_shard_exchange_known per file -> ukm_merge_k of the received slices), the leg bench.py --gpus N times beside the
    torch.distributed one.  One GPU allows a one-rank communicator only: the calls, the count tables and the rebuild run;
    the result is the input."""
    import torch
    from unikmer_amd import dist as ud
    from unikmer_amd import lib
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ctx.comm_init(1, 0, lib.Context.comm_unique_id())
    rng = np.random.default_rng(9)
    files = [np.unique(rng.integers(0, 1 << 62, n, dtype=np.uint64)) for n in (200_000, 1, 50_000)]
    taxs = [rng.integers(1, 999, len(f)).astype(np.uint32) for f in files]
    tf = [torch.from_numpy(f.view(np.int64)).to(dev) for f in files]
    tt = [torch.from_numpy(t.view(np.int32)).to(dev) for t in taxs]
    local, local_t = ud.redistribute_cabi(ctx, tf, 62, tt)
    for f, t, lk, lt in zip(files, taxs, local, local_t):
        assert np.array_equal(lk.cpu().numpy().view(np.uint64), f) and np.array_equal(lt.cpu().numpy().view(np.uint32), t)
    local, none = ud.redistribute_cabi(ctx, tf, 62)
    assert none is None and all(np.array_equal(lk.cpu().numpy().view(np.uint64), f) for f, lk in zip(files, local))
    ctx.comm_destroy()
    ctx.close()
