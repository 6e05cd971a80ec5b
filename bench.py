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

N > 1: the code space is sharded by high-bits prefix.  `value` is the job END TO END: every rank starts from a
FILE-sharded state (a stride sample of both global sets), cuts both at the prefix splitters, exchanges each set ONCE
(RCCL all-to-all-v over xGMI, unikmer_amd/dist.py), k-way merges what arrived and runs union + inter on its range --
that figure is bounded by xGMI, not HBM.  `value_prepartitioned` is the same job on inputs that
already sit on their range owners (no data-path collective); `value_offset_sharded_start` the end-to-end job from an
OFFSET-sharded start (rank r holds the r-th contiguous chunk of each sorted set: the slices arrive in value order and are
not merged); `exchange` is one bare all-to-all-v.  --scaling weak (default):
--set-size k-mers per set PER GPU; --scaling strong: in total (the metric's wording).
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
                  "reference's hash-map union and 2-pointer inter, 1 thread; Go toolchain absent; the survey's sizes 2 x 1e8 "
                  "and 2 x 1.25e8 were run once: profiles/r06_cpu_baseline_sizes.json, 2.51e7 / 2.24e7 k-mers/s)" % len(A),
        "union_s": tu, "inter_s": ti, "union_out": int(len(u)), "inter_out": int(len(i)),
        "host_cores_available": os.cpu_count(),
        "allcores_sorted_merge": {"value": kmers / (best[0] + best[1]), "unit": "k-mers/s", "cores": best[2],
                                  "union_s": best[0], "inter_s": best[1], "reps": 6, "statistic": "per-op minimum",
                                  "note": "not the reference's algorithm: value-range partitioned 2-pointer merges "
                                          "on every core (count pass + write pass)"},
    }


def _xor_fold(t):
    """XOR checksum of an int64 device tensor"""
    import torch
    x = t
    while x.numel() > 1:
        if x.numel() & 1:
            x = torch.cat([x, torch.zeros(1, dtype=x.dtype, device=x.device)])
        h = x.numel() // 2
        x = x[:h] ^ x[h:]
    return int(x.item()) if x.numel() else 0


def check_outputs(A, B, U, I, window=1_000_000):
    """Full-size parity properties of one union + inter result (run once, on the warm-up outputs; tests/ hold the
    bit-exact comparisons with the oracle): inclusion-exclusion, strict order, the XOR checksum of checksums
    xor(U) = xor(A) ^ xor(B) ^ xor(I), and -- on windows cut from both ends and the middle of each output -- equality
    with numpy's set operation on the matching input slices plus the window's rank derived from the inputs."""
    import torch
    na, nb, nu, ni = A.numel(), B.numel(), U.numel(), I.numel()
    assert nu + ni == na + nb, "inclusion-exclusion violated"
    assert bool((U[1:] > U[:-1]).all()) and bool((I[1:] > I[:-1]).all()), "output not strictly sorted"
    assert _xor_fold(U) == _xor_fold(A) ^ _xor_fold(B) ^ _xor_fold(I), "XOR checksum of checksums violated"

    def lower(S, v):
        return int(torch.searchsorted(S, torch.tensor([v], dtype=torch.int64, device=S.device)).item())

    def host(t):
        return t.cpu().numpy().view(np.uint64)
    for out, other, fn in ((U, I, np.union1d), (I, U, np.intersect1d)):
        n = out.numel()
        w = min(window, n)
        for start in sorted({0, max(0, n // 2 - w // 2), n - w}):
            if w == 0:
                continue
            win = out[start:start + w]
            lo, hi = int(win[0].item()), int(win[-1].item())
            a0, a1, b0, b1 = lower(A, lo), lower(A, hi + 1), lower(B, lo), lower(B, hi + 1)
            assert np.array_equal(host(win), fn(host(A[a0:a1]), host(B[b0:b1]))), "window at %d differs from numpy" % start
            assert start == a0 + b0 - lower(other, lo), "window at %d sits at the wrong rank" % start


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--set-size", dest="n", type=float, default=1e9,
                    help="k-mers per set: per GPU with --scaling weak, in total with --scaling strong")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --set-size k-mers per set PER GPU (default); strong = in total (what the metric words)")
    ap.add_argument("--cpu-sample", type=float, default=2e7, help="k-mers per set for the CPU baseline (0 = skip)")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: skip the end-to-end leg; `value` is then the pre-partitioned figure (and says so)")
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
    cdev = torch.device("cpu") if one_gpu else dev

    n = int(args.n) if args.scaling == "weak" else max(1, int(args.n) // world)   # k-mers per set on this rank
    n_universe = (4 * n + 2) // 3
    log2w = max(0, (world - 1).bit_length())
    gap_bits = 32 - log2w                      # keeps world * sum(gaps) below 2^62 (k=31)
    base = rank * ((1 << 62) // world)         # prefix shard r of the k=31 code space
    while n_universe * (1 << (gap_bits - 1)) > (1 << 62) // max(world, 1) and gap_bits > 2:
        gap_bits -= 1
    A, B = gen_sets_device(n_universe, gap_bits, base, SEED + 7919 * rank, dev)
    na, nb = A.numel(), B.numel()

    # verify the device generator against numpy on a 1e6 sub-sample (rank's own seed/base)
    An, Bn = gen_sets_numpy(min(1_000_000, n_universe), gap_bits, base, SEED + 7919 * rank)
    ma, mb = min(na, len(An)), min(nb, len(Bn))
    assert np.array_equal(A[:ma].cpu().numpy().view(np.uint64), An[:ma]), "device generator != numpy generator"
    assert np.array_equal(B[:mb].cpu().numpy().view(np.uint64), Bn[:mb]), "device generator != numpy generator"

    # output buffers are resident (first-touched) before anything is timed, like the inputs
    out_u = torch.zeros(na + nb, dtype=torch.int64, device=dev)
    out_i = torch.zeros(min(na, nb), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx = lib.Context(dev.index, stream=stream.cuda_stream)

    def step():
        u = ctx.setop2(lib.OP_UNION, A, B, out=out_u)
        ku = ctx.last_kernel_ms()
        i = ctx.setop2(lib.OP_INTER, A, B, out=out_i)
        ki = ctx.last_kernel_ms()
        return u.numel(), i.numel(), ku, ki

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(vals):
        if world == 1:
            return [int(v) for v in vals]
        t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(x) for x in t.cpu()]

    # ---- the single-GPU path on this rank's prefix range (N = 1: the whole job) -------------------------------
    for _ in range(max(1, args.warmup)):
        nu, ni, _, _ = step()
    check_outputs(A, B, out_u[:nu], out_i[:ni])   # full-size parity properties, once, outside the timed region
    step()  # untimed: the check's multi-GB temporaries have just been freed; the first step behind them runs ~10 % slow
    barrier()
    t0 = time.perf_counter()
    ku_sum = ki_sum = 0.0
    for _ in range(args.steps):
        nu, ni, ku, ki = step()
        ku_sum += ku
        ki_sum += ki
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    tot_in, tot_u, tot_i = sum_over_ranks([na + nb, nu, ni])
    ms_pre = dt * 1e3 / args.steps
    value_pre = 2.0 * tot_in * args.steps / dt  # each op consumes |A|+|B| input k-mers

    # ---- taxid variant (OUTSIDE the timed region; rank 0): the same two sets with taxids, so that a driver-run number exists
    # for the paths north_star names ("with per-k-mer TaxId LCA reduction").  (a) per-RECORD taxids, SURVEY 8(d)'s generator:
    # taxid = 1 + splitmix64(seed3 ^ code) mod T on the complete 8-ary tree of depth 7 (uniformly random: the adversarial
    # case, every match is an LCA of two unrelated leaves); (b) ONE taxid per file (the reference's documented workflow:
    # `count -t`, the .unik header's global taxid) handed over as a scalar (ukm_setop2_ft).  Kernel ms by hipEvents.
    taxid_variant = None
    if rank == 0 and os.environ.get("UKM_BENCH_NO_TAXID") != "1":
        try:
            T = sum(8 ** d for d in range(8))
            child = np.arange(1, T + 1, dtype=np.uint32)
            parent = ((child.astype(np.int64) - 2) // 8 + 1).astype(np.uint32)
            parent[0] = 1
            ctx.taxonomy_load(child, parent)
            tout_u = torch.zeros(na + nb, dtype=torch.int32, device=dev)
            tout_i = torch.zeros(min(na, nb), dtype=torch.int32, device=dev)

            def kms(fn, reps=3):
                best = None
                for _ in range(reps + 1):
                    r = fn()
                    k = ctx.last_kernel_ms()
                    best = k if best is None else min(best, k)
                return best, r
            leaves = T - 8 ** 7 + 1
            fa, fb = int(leaves + 5), int(leaves + 8 ** 6 + 77)  # two leaves under different children of the root
            lab = int(ctx.lca(np.array([fa], dtype=np.uint32), np.array([fb], dtype=np.uint32))[0])
            tu, ru = kms(lambda: ctx.setop2(lib.OP_UNION, A, B, fa, fb, out=out_u, out_taxids=tout_u))
            ti, ri = kms(lambda: ctx.setop2(lib.OP_INTER, A, B, fa, fb, out=out_i, out_taxids=tout_i))
            assert ru[0].numel() == nu and ri[0].numel() == ni
            # the taxids of the union: A's own, B's own, or the root on the codes both hold -- counted at full size
            cu = ru[1]
            n_root, n_a, n_b = int((cu == lab).sum().item()), int((cu == fa).sum().item()), int((cu == fb).sum().item())
            assert lab == 1 and (n_root, n_a, n_b) == (ni, na - ni, nb - ni), "per-file taxids of the union are wrong"
            assert bool((ri[1] == lab).all()), "per-file taxids of the intersection are wrong"
            per_file = {"union_kernel_ms": tu, "inter_kernel_ms": ti,
                        "union_frac": (8.0 * (na + nb) + 12.0 * nu) / (tu * 1e-3) / 1e9 / 8000.0,
                        "inter_frac": (8.0 * (na + nb) + 12.0 * ni) / (ti * 1e-3) / 1e9 / 8000.0,
                        "algorithmic_bytes": "8 B per input record (no taxid is read: one value per file), 12 B per output record",
                        "checked": "taxid histogram of the full-size outputs (A's / B's / LCA = root)"}
            del cu
            # (SURVEY 8(d) words the generator as a function of the code alone; with ONE seed both files would give a shared
            #  code the same taxid and every LCA would be the trivial LCA(x, x) -- the two files get their own seeds instead)
            ta = (1 + (splitmix64_torch(A ^ _i64(SEED + 2)) & ((1 << 40) - 1)) % T).to(torch.int32)
            tb = (1 + (splitmix64_torch(B ^ _i64(SEED + 3)) & ((1 << 40) - 1)) % T).to(torch.int32)
            tu2, ru2 = kms(lambda: ctx.setop2(lib.OP_UNION, A, B, ta, tb, out=out_u, out_taxids=tout_u))
            ti2, ri2 = kms(lambda: ctx.setop2(lib.OP_INTER, A, B, ta, tb, out=out_i, out_taxids=tout_i))
            assert ru2[0].numel() == nu and ri2[0].numel() == ni
            # a window of the intersection against the bulk LCA entry point on the taxids of the matching input records
            w = min(2_000_000, ni)
            ia, ib = torch.searchsorted(A, ri2[0][:w]), torch.searchsorted(B, ri2[0][:w])
            exp = ctx.lca(ta[ia].contiguous(), tb[ib].contiguous())
            assert torch.equal(ri2[1][:w], exp), "per-record taxids of the intersection are wrong"
            del ia, ib
            per_record = {"union_kernel_ms": tu2, "inter_kernel_ms": ti2,
                          "union_frac": (12.0 * (na + nb) + 12.0 * nu) / (tu2 * 1e-3) / 1e9 / 8000.0,
                          "inter_frac": (12.0 * (na + nb) + 12.0 * ni) / (ti2 * 1e-3) / 1e9 / 8000.0,
                          "algorithmic_bytes": "12 B per input and output record (u64 code + u32 taxid)",
                          "checked": "a sanity check of the bench leg, NOT parity: output sizes, and one window of %d intersection "
                                     "records against the library's own bulk LCA entry point (ukm_lca) on the matching input taxids; "
                                     "the per-record union is size-checked only.  Parity of this kernel against the oracle: "
                                     "tests/test_gpu_parity.py (taxid set operations), tests/test_gpu_filetax.py" % w,
                          "generator": "taxid = 1 + splitmix64(seed ^ code) mod T with one seed per file, complete 8-ary tree of depth 7 "
                                       "(SURVEY 8(d)): uniformly random, every match is an LCA of two unrelated nodes",
                          "second_bound": "beside the bytes, every match costs two random one-byte reads of the 2.4 MB clade table: "
                                          "%.2e of them per operation; torch's own gather reads that table at 194 G reads/s "
                                          "(tools/gather_ceiling.py, profiles/r06_notes.md section 10)" % (2.0 * ni)}
            taxid_variant = {"outside_timed_region": True, "set_size": n, "one_taxid_per_file": per_file, "per_record_taxids": per_record,
                             "unit": "kernel ms by hipEvents (best of 4); frac = algorithmic bytes / kernel time / 8 TB/s"}
            del ta, tb, tout_u, tout_i, exp
        except Exception as e:  # never in the way of the headline
            taxid_variant = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    res = None
    if rank == 0:
        # roofline of the dominant kernel (union tile kernel): algorithmic bytes per launch
        # = 8(|A|+|B|) read + 8|A∪B| written, rank 0's launch
        ku = ku_sum / args.steps * 1e-3
        ki = ki_sum / args.steps * 1e-3
        bytes_u = 8.0 * (na + nb) + 8.0 * nu
        bytes_i = 8.0 * (na + nb) + 8.0 * ni
        peak = 8000.0
        # HBM traffic is a PMC measurement and cannot be taken inside this process: the figure below was collected by
        # tools/collect_profiles.sh (separate rocprofv3 --pmc passes over THIS command) and is committed with the head
        # it was measured at; it describes the kernel, not this particular run
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if int(tj.get("n", 0)) == n:
                    traffic = tj.get("union_traffic_bytes_per_launch")
                    traffic_source = {"file": "profiles/traffic.json", "collected_at_head": tj.get("head"),
                                      "method": tj.get("method", "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, own passes"),
                                      "note": "profile-derived: measured on the same command in a separate rocprofv3 run, not in this run"}
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "setop_tile_kernel<UNION>", "achieved": bytes_u / ku / 1e9,
                    "peak": peak, "unit": "GB/s", "frac": bytes_u / ku / 1e9 / peak, "traffic": traffic,
                    "traffic_from_profile": traffic_source,
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
            "value": value_pre, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_pre, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "union+inter of two sorted k=31 sets of %d uint64 codes per GPU "
                                   "(|A|=%d |B|=%d |A∪B|=%d |A∩B|=%d on rank 0)" % (n, na, nb, nu, ni),
                       "k": 31, "per_gpu_set_size": n, "global_set_size": n * world,
                       "parallelism": "prefix-sharded x%d" % world,
                       "ops_per_step": ["ukm_setop2(UNION)", "ukm_setop2(INTER)"]},
            "roofline": roofline, "roofline_inter": roofline_inter, "cpu_baseline": cpu, "taxid_variant": taxid_variant,
            "union_kmers_per_s_kernel": (na + nb) / ku, "inter_kmers_per_s_kernel": (na + nb) / ki,
            "value_prepartitioned": value_pre, "ms_per_step_prepartitioned": ms_pre,
            "parity_checked": "inclusion-exclusion, strict order, XOR checksum of checksums, 6 numpy-checked windows of 1e6 "
                              "records with their ranks (warm-up outputs, full size)",
        }
    # The leg below talks to the other ranks.  If one of them fails there (out of memory, an RCCL error) the
    # others would wait in a collective for ever and the numbers measured above would be lost with them: a
    # watchdog prints the line without the leg and ends the process instead.
    import threading
    leg_done = threading.Event()

    def _bail():
        if leg_done.is_set():
            return
        if rank == 0 and res is not None:
            res["value_is"] = "PRE-PARTITIONED (no redistribution): the end-to-end leg did not finish within %d s" % LEG_TIMEOUT
            res["exchange"] = {"error": "timeout"}
            print(json.dumps(res), flush=True)
        os._exit(0)
    LEG_TIMEOUT = int(os.environ.get("UKM_BENCH_LEG_TIMEOUT", "300"))
    watchdog = None
    if world > 1 and not args.no_exchange:
        watchdog = threading.Timer(LEG_TIMEOUT, _bail)
        watchdog.daemon = True
        watchdog.start()
    # ---- N > 1: the job END TO END from a file-sharded start (north_star's path includes the redistribution) ----
    # Every rank holds a 1/world stride sample of the GLOBAL A and of the global B (sorted, spanning the whole code
    # space: what a rank has after reading its share of the input files).  One step = cut both sets at the prefix
    # splitters, ONE all-to-all-v per set over RCCL/xGMI (dist.redistribute: each input travels once and serves both
    # operations), k-way merge of the received slices, then union + inter through the 2-way kernel on the rank's
    # range.  This is `value` at N > 1; the same job without the exchange is `value_prepartitioned`.
    exchange = None
    incl = None
    if world > 1 and not args.no_exchange:
        try:
            from unikmer_amd import dist as ud

            def file_shard(X):
                send = torch.cat([X[r::world] for r in range(world)])
                counts = [X[r::world].numel() for r in range(world)]
                full, _, _ = ud.exchange_sorted(send, counts)      # world sorted pieces in range order = sorted
                return full
            Af, Bf = file_shard(A), file_shard(B)
            spl = ud.prefix_splitters(62, world)[:-1]

            def step_e2e():
                (Al, Bl), _ = ud.redistribute(ctx, [Af, Bf], 62)
                u = ctx.setop2(lib.OP_UNION, Al, Bl, out=out_u)
                i = ctx.setop2(lib.OP_INTER, Al, Bl, out=out_i)
                return Al, Bl, u, i
            for _ in range(max(1, min(args.warmup, 2))):            # workspace, RCCL channels
                Al, Bl, eu, ei = step_e2e()
            assert torch.equal(Al, A) and torch.equal(Bl, B), "redistribution did not rebuild the rank's range"
            assert eu.numel() == nu and ei.numel() == ni
            del Al, Bl
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                _, _, eu, ei = step_e2e()
            barrier()
            t1 = max_over_ranks(time.perf_counter() - t1)
            g_in, g_u, g_i = sum_over_ranks([Af.numel() + Bf.numel(), eu.numel(), ei.numel()])
            assert g_u + g_i == g_in, "inclusion-exclusion violated in the end-to-end run"
            assert (g_u, g_i) == (tot_u, tot_i), "end-to-end result sizes differ from the pre-partitioned run"
            incl = {"value": 2.0 * g_in * args.steps / t1, "unit": "k-mers/s", "steps": args.steps,
                    "ms_per_step": t1 * 1e3 / args.steps,
                    "note": "file-sharded start -> cut at the prefix splitters -> one all-to-all-v per input set -> k-way merge "
                            "of the received slices -> union + inter on the rank's range (each input is exchanged once per step)"}
            # The same job from an OFFSET-sharded start (rank r holds the r-th contiguous chunk of each globally sorted
            # set: what a rank has after reading its share of one sorted file, SURVEY 8(e)'s example).  The slices that
            # arrive are already in value order, so dist.redistribute hands back the receive buffer and no merge pass
            # runs; with uniform synthetic codes nearly everything already sits on its range owner, so this leg shows the
            # floor of the end-to-end step (cuts + a near-empty all-to-all-v + union + inter).
            try:
                def offset_shard(X):
                    sizes = [0] * world
                    sizes[rank] = X.numel()
                    tot = sum_over_ranks(sizes)                     # every rank's range size
                    first, N = sum(tot[:rank]), sum(tot)
                    counts = []
                    for r in range(world):                          # records of mine inside chunk r = [r N / W, (r + 1) N / W)
                        lo_r, hi_r = r * N // world, (r + 1) * N // world
                        counts.append(max(0, min(hi_r, first + X.numel()) - max(lo_r, first)))
                    full, _, _ = ud.exchange_sorted(X, counts)
                    return full
                Ao, Bo = offset_shard(A), offset_shard(B)

                def step_off():
                    (Al, Bl), _ = ud.redistribute(ctx, [Ao, Bo], 62)
                    u = ctx.setop2(lib.OP_UNION, Al, Bl, out=out_u)
                    i = ctx.setop2(lib.OP_INTER, Al, Bl, out=out_i)
                    return Al, Bl, u, i
                Al, Bl, ou, oi = step_off()
                assert torch.equal(Al, A) and torch.equal(Bl, B), "offset-sharded redistribution did not rebuild the rank's range"
                assert ou.numel() == nu and oi.numel() == ni
                del Al, Bl
                barrier()
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    step_off()
                barrier()
                t2 = max_over_ranks(time.perf_counter() - t2)
                incl["offset_sharded_start"] = {
                    "value": 2.0 * g_in * args.steps / t2, "unit": "k-mers/s", "ms_per_step": t2 * 1e3 / args.steps,
                    "note": "rank r starts with the r-th contiguous chunk of each sorted set: the received slices are in "
                            "value order and are used as they lie in the receive buffer (no merge pass)"}
                del Ao, Bo
            except Exception as e:
                incl["offset_sharded_start"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            # The same step with each rank's range exchanged in FOUR sub-ranges by value: the rebuild of the slices that have
            # arrived overlaps the transfer of the next sub-range (dist.redistribute_pipelined)
            try:
                def step_pipe():
                    (Al, Bl), _ = ud.redistribute(ctx, [Af, Bf], 62, pipeline=4)
                    u = ctx.setop2(lib.OP_UNION, Al, Bl, out=out_u)
                    i = ctx.setop2(lib.OP_INTER, Al, Bl, out=out_i)
                    return Al, Bl, u, i
                Al, Bl, pu, pi = step_pipe()
                assert torch.equal(Al, A) and torch.equal(Bl, B), "pipelined redistribution did not rebuild the rank's range"
                assert pu.numel() == nu and pi.numel() == ni
                del Al, Bl
                barrier()
                t4 = time.perf_counter()
                for _ in range(args.steps):
                    step_pipe()
                barrier()
                t4 = max_over_ranks(time.perf_counter() - t4)
                incl["pipelined_4_subranges"] = {
                    "value": 2.0 * g_in * args.steps / t4, "unit": "k-mers/s", "ms_per_step": t4 * 1e3 / args.steps,
                    "note": "the end-to-end step with every input exchanged sub-range by sub-range (4 per rank): transfers overlap rebuilds"}
            except Exception as e:
                incl["pipelined_4_subranges"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            # The same end-to-end step through the LIBRARY'S OWN exchange -- ukm_comm_init + ukm_shard_counts +
            # ukm_shard_exchange_known (grouped ncclSend / ncclRecv in ukm_comm.hip), the calls INTEGRATION.md's Go host
            # makes; torch.distributed only hands the communicator id over.  RCCL wants one device per rank, so the 1-GPU
            # test hook skips it.
            if one_gpu:
                incl["exchange_cabi"] = {"skipped": "UKM_BENCH_ONE_GPU: RCCL does not take two ranks on one device"}
            else:
                try:
                    ud.comm_init_from_dist(ctx)

                    def step_cabi():
                        (Al, Bl), _ = ud.redistribute_cabi(ctx, [Af, Bf], 62)
                        u = ctx.setop2(lib.OP_UNION, Al, Bl, out=out_u)
                        i = ctx.setop2(lib.OP_INTER, Al, Bl, out=out_i)
                        return Al, Bl, u, i
                    Al, Bl, cu, ci = step_cabi()
                    assert torch.equal(Al, A) and torch.equal(Bl, B), "C-ABI redistribution did not rebuild the rank's range"
                    assert cu.numel() == nu and ci.numel() == ni
                    del Al, Bl
                    barrier()
                    t3 = time.perf_counter()
                    for _ in range(args.steps):
                        step_cabi()
                    barrier()
                    t3 = max_over_ranks(time.perf_counter() - t3)
                    incl["exchange_cabi"] = {
                        "value": 2.0 * g_in * args.steps / t3, "unit": "k-mers/s", "ms_per_step": t3 * 1e3 / args.steps,
                        "note": "the end-to-end step with the exchange behind the C ABI (ukm_shard_counts + "
                                "ukm_shard_exchange_known per set, ukm_merge_k of the received slices): what a Go host runs"}
                    ctx.comm_destroy()
                except Exception as e:
                    incl["exchange_cabi"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            # the bare all-to-all-v of one set, for the link rate
            counts = ud.cuts_to_counts(ctx.partition_points(Af, spl), Af.numel())
            barrier()
            te = time.perf_counter()
            reps = 3
            for _ in range(reps):
                got, _, _ = ud.exchange_sorted(Af, counts)
            barrier()
            te = max_over_ranks((time.perf_counter() - te) / reps)
            moved = Af.numel() - counts[rank]
            exchange = {"ms": te * 1e3, "records_per_rank": int(Af.numel()), "bytes_sent_per_rank": int(moved) * 8,
                        "GBps_per_rank": int(moved) * 8 / te / 1e9,
                        "note": "one all-to-all-v (RCCL) redistributing one file-sharded set of ~n codes per rank "
                                "to its prefix owners"}
            del got, Af, Bf
        except Exception as e:  # the pre-partitioned numbers above never depend on this leg
            exchange = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    leg_done.set()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if exchange:
            res["exchange"] = exchange
        if world > 1:
            if incl:
                # the headline at N > 1 is the end-to-end figure
                res["value"] = incl["value"]
                res["ms_per_step"] = incl["ms_per_step"]
                res["value_is"] = "end to end, including the prefix redistribution (all-to-all-v) of both inputs in every step"
                res["value_incl_exchange"] = incl["value"]
                res["incl_exchange"] = incl
                if "value" in incl.get("offset_sharded_start", {}):
                    res["value_offset_sharded_start"] = incl["offset_sharded_start"]["value"]
                if "value" in incl.get("pipelined_4_subranges", {}):
                    res["value_pipelined_exchange"] = incl["pipelined_4_subranges"]["value"]
                if "value" in incl.get("exchange_cabi", {}):
                    res["value_exchange_cabi"] = incl["exchange_cabi"]["value"]
            else:
                res["value_is"] = ("PRE-PARTITIONED (no redistribution in the timed region): " +
                                   ("--no-exchange was given" if args.no_exchange else "the end-to-end leg failed, see `exchange`"))
        if cpu:
            res["speedup_vs_cpu_port"] = res["value"] / cpu["value"]
            res["speedup_vs_cpu_allcores_merge"] = res["value"] / cpu["allcores_sorted_merge"]["value"]
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
