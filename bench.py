#!/usr/bin/env python
"""bench.py — k-mers/sec for `union` + `inter` of two sorted k=31 k-mer sets on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
A step = one pass of the hot path over one batch = ukm_setop2(UNION) + ukm_setop2(INTER) on
the two device-resident sets, through the C ABI of libunikmer_hip.so.  Inputs (and output
buffers) are resident in HBM when the timed region starts; per step every rank processes
2 x (|A|+|B|) input k-mers.

Workload (BASELINE.json metric): synthetic sets of --n (default 1e9) uint64
k=31 codes each, generated on the device: universe U[i] = prefix sum of gaps
1 + (splitmix64(seed ^ j) mod G), membership m = splitmix64(seed2 ^ i) & 3 (0 -> A only,
1 -> B only, 2/3 -> both), |U| = 4n/3, so |A ∩ B| ≈ 2n/3 and |A ∪ B| = 4n/3.

N > 1 (weak scaling): the code space is sharded by high-bits prefix; rank r holds the r-th
prefix range of both sets (|A_r| ≈ |B_r| ≈ n), i.e. the state after the prefix redistribution,
and runs the 1-GPU path on it with no data-path collective.  The redistribution itself
(RCCL all-to-all-v over xGMI, unikmer_amd/dist.py) is reported twice: "exchange" = one bare all-to-all-v, and
"value_incl_exchange" = the same union + inter job end to end from a FILE-sharded start (cut, exchange, merge of
the received pieces, 2-way op) — that figure is bounded by xGMI, not HBM.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x756E696B6D6572  # "unikmer"
M64 = (1 << 64) - 1


def _i64(x):
    """python int (uint64 value) -> the int64 with the same bit pattern"""
    x &= M64
    return x - (1 << 64) if x >= (1 << 63) else x


def splitmix64_torch(x):
    """splitmix64 finaliser on int64 tensors holding uint64 bit patterns (logical shifts)."""
    import torch
    z = x + _i64(0x9E3779B97F4A7C15)
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _i64(0x94D049BB133111EB)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def splitmix64_np(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gen_sets_device(n_universe, gap_bits, base, seed, device, chunk=1 << 27):
    """Returns (A, B) int64 device tensors (uint64 bit patterns), sorted, strictly increasing."""
    import torch
    A_parts, B_parts = [], []
    run = base
    for lo in range(0, n_universe, chunk):
        hi = min(lo + chunk, n_universe)
        j = torch.arange(lo, hi, dtype=torch.int64, device=device)
        gaps = 1 + (splitmix64_torch(j ^ _i64(seed)) & ((1 << gap_bits) - 1))
        U = torch.cumsum(gaps, 0) + run
        run = int(U[-1].item())
        m = splitmix64_torch(j ^ _i64(seed + 1)) & 3
        A_parts.append(U[(m == 0) | (m >= 2)])
        B_parts.append(U[(m == 1) | (m >= 2)])
        del j, gaps, U, m
    A = torch.cat(A_parts)
    B = torch.cat(B_parts)
    del A_parts, B_parts
    torch.cuda.empty_cache()
    return A, B


def gen_sets_numpy(n_universe, gap_bits, base, seed):
    j = np.arange(n_universe, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64_np(j ^ np.uint64(seed)) & np.uint64((1 << gap_bits) - 1))
    U = np.cumsum(gaps, dtype=np.uint64) + np.uint64(base)
    m = splitmix64_np(j ^ np.uint64(seed + 1)) & np.uint64(3)
    return U[(m == 0) | (m >= 2)], U[(m == 1) | (m >= 2)]


def cpu_baseline(sample_universe, gap_bits):
    """The reference's algorithms (hash-map union + sort of keys, union.go:186-305; 2-pointer
    inter, inter.go:205-278) as restated in oracle/, single thread, on a bounded sample of the
    same generator.  Reported, never the target."""
    from oracle import oracle as O
    A, B = gen_sets_numpy(sample_universe, gap_bits, 0, SEED)
    tu, u = O.time_union2(A, B)
    ti, i = O.time_inter2(A, B)
    kmers = 2 * (len(A) + len(B))
    # the best a CPU does with every core (parallel sorted merge), beside the
    # reference-equivalent single-thread figure.  Per-op MINIMUM over 6 repetitions: single runs of a 128-thread
    # merge on a shared host vary by 10x (page faults, other tenants); the minimum is the stable figure.
    pus, pis, th = [], [], 1
    for _ in range(6):
        pu, pu_out, th = O.time_setop2_allcores(0, A, B)
        pi, pi_out, _ = O.time_setop2_allcores(1, A, B)
        pus.append(pu)
        pis.append(pi)
    best = (min(pus), min(pis), th)
    assert len(pu_out) == len(u) and len(pi_out) == len(i)
    return {
        "value": kmers / (tu + ti),
        "unit": "k-mers/s",
        "cores": 1,
        "kind": "port",
        "sample": "union+inter of 2 x %d synthetic k=31 codes (same generator; C restatement of the "
                  "reference's hash-map union and 2-pointer inter, 1 thread; Go toolchain absent)" % len(A),
        "union_s": tu, "inter_s": ti, "union_out": int(len(u)), "inter_out": int(len(i)),
        "host_cores_available": os.cpu_count(),
        "allcores_sorted_merge": {"value": kmers / (best[0] + best[1]), "unit": "k-mers/s", "cores": best[2],
                                  "union_s": best[0], "inter_s": best[1], "reps": 6, "statistic": "per-op minimum",
                                  "note": "not the reference's algorithm: value-range partitioned 2-pointer merges "
                                          "on every core (count pass + write pass)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--set-size", dest="n", type=float, default=1e9, help="k-mers per set per GPU")
    ap.add_argument("--cpu-sample", type=float, default=2e7, help="k-mers per set for the CPU baseline (0 = skip)")
    ap.add_argument("--no-exchange", action="store_true", help="skip the separate all-to-all timing at N>1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from unikmer_amd import lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # UKM_BENCH_ONE_GPU=1 is a TEST HOOK for 1-GPU boxes: all ranks share cuda:0 and the control-plane
    # collectives go over gloo, so the N>1 code path (sharded generation, max-over-ranks timing, sums)
    # can be exercised without a multi-GPU node (device tensors are staged through the host for the gloo all-to-all).
    # Its numbers mean nothing.
    one_gpu = world > 1 and os.environ.get("UKM_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        import datetime
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10),
                                    device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    n = int(args.n)
    n_universe = (4 * n + 2) // 3
    log2w = max(0, (world - 1).bit_length())
    gap_bits = 32 - log2w                      # keeps world * sum(gaps) below 2^62 (k=31)
    base = rank * ((1 << 62) // world)         # prefix shard r of the k=31 code space
    while n_universe * (1 << (gap_bits - 1)) > (1 << 62) // max(world, 1) and gap_bits > 2:
        gap_bits -= 1
    A, B = gen_sets_device(n_universe, gap_bits, base, SEED + 7919 * rank, dev)
    na, nb = A.numel(), B.numel()

    # verify the device generator against numpy on a 1e6 sub-sample (rank's own seed/base)
    An, Bn = gen_sets_numpy(1_000_000, gap_bits, base, SEED + 7919 * rank)
    ma, mb = min(na, len(An)), min(nb, len(Bn))
    assert np.array_equal(A[:ma].cpu().numpy().view(np.uint64), An[:ma]), "device generator != numpy generator"
    assert np.array_equal(B[:mb].cpu().numpy().view(np.uint64), Bn[:mb]), "device generator != numpy generator"

    out_u = torch.empty(na + nb, dtype=torch.int64, device=dev)
    out_i = torch.empty(min(na, nb), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx = lib.Context(dev.index, stream=stream.cuda_stream)

    def step():
        u = ctx.setop2(lib.OP_UNION, A, B, out=out_u)
        ku = ctx.last_kernel_ms()
        i = ctx.setop2(lib.OP_INTER, A, B, out=out_i)
        ki = ctx.last_kernel_ms()
        return u.numel(), i.numel(), ku, ki

    for _ in range(args.warmup):
        nu, ni, _, _ = step()
    if args.warmup == 0:
        nu, ni, _, _ = step()
    # size-independent parity properties at full size (tests/ hold the bit-exact comparisons)
    assert nu + ni == na + nb, "inclusion-exclusion violated"
    u_t, i_t = out_u[:nu], out_i[:ni]
    assert bool((u_t[1:] > u_t[:-1]).all()) and bool((i_t[1:] > i_t[:-1]).all()), "output not strictly sorted"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    ku_sum = ki_sum = 0.0
    for _ in range(args.steps):
        nu, ni, ku, ki = step()
        ku_sum += ku
        ki_sum += ki
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        cdev = torch.device("cpu") if one_gpu else dev
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([na + nb, nu, ni], dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_in, tot_u, tot_i = (int(x) for x in cnt.cpu())
    else:
        tot_in, tot_u, tot_i = na + nb, nu, ni

    ms_per_step = dt * 1e3 / args.steps
    value = 2.0 * tot_in * args.steps / dt  # each op consumes |A|+|B| input k-mers

    res = None
    if rank == 0:
        # roofline of the dominant kernel (union tile kernel): algorithmic bytes per launch
        # = 8(|A|+|B|) read + 8|A∪B| written, rank 0's launch
        ku = ku_sum / args.steps * 1e-3
        ki = ki_sum / args.steps * 1e-3
        bytes_u = 8.0 * (na + nb) + 8.0 * nu
        bytes_i = 8.0 * (na + nb) + 8.0 * ni
        peak = 8000.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if int(tj.get("n", 0)) == n:
                    traffic = tj.get("union_traffic_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "setop_tile_kernel<UNION>", "achieved": bytes_u / ku / 1e9,
                    "peak": peak, "unit": "GB/s", "frac": bytes_u / ku / 1e9 / peak, "traffic": traffic,
                    "algorithmic_bytes": bytes_u, "kernel_ms": ku * 1e3,
                    # SURVEY §8(d) also asks for the read side alone (8(|A|+|B|) bytes over the same time)
                    "read_only_achieved": 8 * (na + nb) / ku / 1e9, "read_only_frac": 8 * (na + nb) / ku / 1e9 / peak,
                    "note": "frac = (8(|A|+|B|) read + 8|out| written) / kernel time / 8 TB/s.  north_star's '>= 50 % of "
                            "HBM-read roofline' is read_only_frac >= 0.5, i.e. kernel <= 4.0 ms at this size: reachable for "
                            "inter (21.3 GB total), not for union, whose 10.7 GB of output make 4.0 ms = 6.7 TB/s of mixed "
                            "traffic, above the ~6.3 TB/s this part sustains on a plain copy"}
        roofline_inter = {"bound": "hbm", "kernel": "setop_tile_kernel<INTER>", "achieved": bytes_i / ki / 1e9,
                          "peak": peak, "unit": "GB/s", "frac": bytes_i / ki / 1e9 / peak,
                          "algorithmic_bytes": bytes_i, "kernel_ms": ki * 1e3,
                          "read_only_achieved": 8 * (na + nb) / ki / 1e9, "read_only_frac": 8 * (na + nb) / ki / 1e9 / peak}
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            cpu = cpu_baseline((4 * int(args.cpu_sample) + 2) // 3, 32)
        res = {
            "metric": "k-mers/sec for union+inter of 1e9-k-mer k=31 sets",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "union+inter of two sorted k=31 sets of %d uint64 codes per GPU "
                                   "(|A|=%d |B|=%d |A∪B|=%d |A∩B|=%d on rank 0)" % (n, na, nb, nu, ni),
                       "k": 31, "per_gpu_set_size": n, "parallelism": "prefix-sharded x%d" % world,
                       "ops_per_step": ["ukm_setop2(UNION)", "ukm_setop2(INTER)"]},
            "roofline": roofline, "roofline_inter": roofline_inter, "cpu_baseline": cpu,
            "union_kmers_per_s_kernel": (na + nb) / ku, "inter_kmers_per_s_kernel": (na + nb) / ki,
        }
    # The leg below talks to the other ranks.  If one of them fails there (out of memory, an RCCL error) the
    # others would wait in a collective for ever and the headline measured above would be lost with them: a
    # watchdog prints the line without the leg and ends the process instead.
    import threading
    leg_done = threading.Event()

    def _bail():
        if leg_done.is_set():
            return
        if rank == 0 and res is not None:
            res["exchange"] = {"error": "the end-to-end leg did not finish within %d s; headline numbers are unaffected" % LEG_TIMEOUT}
            print(json.dumps(res), flush=True)
        os._exit(0)
    LEG_TIMEOUT = int(os.environ.get("UKM_BENCH_LEG_TIMEOUT", "240"))
    watchdog = None
    if world > 1 and not args.no_exchange:
        watchdog = threading.Timer(LEG_TIMEOUT, _bail)
        watchdog.daemon = True
        watchdog.start()
    # ---- N > 1: the same job END TO END from a file-sharded start ("including exchange") ----
    # Every rank holds a 1/world stride sample of the GLOBAL A and of the global B (sorted, spanning the whole code
    # space: what a rank has after reading its share of the input files).  One step = `union` + `inter` through
    # dist.sharded_setop: cut at the prefix splitters, all-to-all-v over RCCL/xGMI, merge of the received pieces,
    # the 2-way kernel on the rank's range.  The pre-partitioned `value` above is the same job without the exchange.
    exchange = None
    incl = None
    if world > 1 and not args.no_exchange:
        try:
            from unikmer_amd import dist as ud
            cdev = torch.device("cpu") if one_gpu else dev

            def file_shard(X):
                send = torch.cat([X[r::world] for r in range(world)])
                counts = [X[r::world].numel() for r in range(world)]
                full, _, _ = ud.exchange_sorted(send, counts)      # world sorted pieces in range order = sorted
                return full
            Af, Bf = file_shard(A), file_shard(B)
            spl = ud.prefix_splitters(62, world)[:-1]
            # (1) the bare all-to-all-v of one set, for the link rate
            counts = ud.cuts_to_counts(ctx.partition_points(Af, spl), Af.numel())
            barrier()
            te = time.perf_counter()
            reps = 3
            for _ in range(reps):
                got, _, _ = ud.exchange_sorted(Af, counts)
            barrier()
            te = (time.perf_counter() - te) / reps
            tt = torch.tensor([te], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            moved = Af.numel() - counts[rank]
            exchange = {"ms": float(tt.item()) * 1e3, "records_per_rank": int(Af.numel()),
                        "bytes_sent_per_rank": int(moved) * 8,
                        "GBps_per_rank": int(moved) * 8 / float(tt.item()) / 1e9,
                        "note": "one all-to-all-v (RCCL) redistributing one file-sharded set of ~n codes per rank "
                                "to its prefix owners"}
            del got
            # (2) union + inter end to end
            esteps = max(1, min(args.steps, 5))
            ud.sharded_setop(ctx, "union", [Af, Bf], 62)           # warm-up (workspace, RCCL channels)
            ud.sharded_setop(ctx, "inter", [Af, Bf], 62)
            barrier()
            t1 = time.perf_counter()
            for _ in range(esteps):
                eu = ud.sharded_setop(ctx, "union", [Af, Bf], 62)
                ei = ud.sharded_setop(ctx, "inter", [Af, Bf], 62)
            barrier()
            t1 = time.perf_counter() - t1
            tt = torch.tensor([t1], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cnt2 = torch.tensor([Af.numel() + Bf.numel(), eu.numel(), ei.numel()], dtype=torch.int64, device=cdev)
            dist.all_reduce(cnt2, op=dist.ReduceOp.SUM)
            g_in, g_u, g_i = (int(x) for x in cnt2.cpu())
            assert g_u + g_i == g_in, "inclusion-exclusion violated in the sharded run"
            assert (g_u, g_i) == (tot_u, tot_i), "sharded result sizes differ from the pre-partitioned run"
            incl = {"value": 2.0 * g_in * esteps / float(tt.item()), "unit": "k-mers/s", "steps": esteps,
                    "ms_per_step": float(tt.item()) * 1e3 / esteps,
                    "note": "file-sharded start -> cut at prefix splitters -> all-to-all-v -> merge of received pieces -> "
                            "2-way op, for union and for inter (each op exchanges its inputs, as two CLI runs would)"}
            del Af, Bf, eu, ei
        except Exception as e:  # the headline numbers above never depend on this leg
            exchange = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    leg_done.set()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if exchange:
            res["exchange"] = exchange
        if incl:
            res["value_incl_exchange"] = incl["value"]
            res["incl_exchange"] = incl
        if cpu:
            res["speedup_vs_cpu_port"] = value / cpu["value"]
            res["speedup_vs_cpu_allcores_merge"] = value / cpu["allcores_sorted_merge"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        # teardown must not hang either (a rank that failed inside the leg is not where the others are)
        t = threading.Timer(60, lambda: os._exit(0))
        t.daemon = True
        t.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
        t.cancel()


if __name__ == "__main__":
    main()
