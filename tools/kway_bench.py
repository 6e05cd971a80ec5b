#!/usr/bin/env python
"""Developer tool: config-3-shaped union (F files x N codes over one universe of 2N, p = 0.5) timed with
UKM_KWAY_DEBUG phase output.  usage: python tools/kway_bench.py [--files 100] [--size 1e8] [--reps 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=100)
    ap.add_argument("--size", type=float, default=1e8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--taxids", action="store_true")
    ap.add_argument("--merge", action="store_true", help="ukm_merge_k PLAIN instead of union")
    a = ap.parse_args()
    import torch
    import bench
    from unikmer_amd import lib
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    per = int(a.size)
    nu = 2 * per
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    gaps = 1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1))
    U = torch.cumsum(gaps, 0)
    del gaps
    files = []
    for f in range(a.files):
        h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
        files.append(U[(h & 1) == 1])
    del j, U
    total = sum(x.numel() for x in files)
    cap = total if a.merge else min(total, nu) + 8
    out = torch.empty(cap, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u = ctx.merge_k(files, out=out) if a.merge else ctx.union(files, out=out)
        torch.cuda.synchronize()
        print("rep %d: %.3f ms  in=%d out=%d" % (r, (time.perf_counter() - t0) * 1e3, total, u.numel()), flush=True)


if __name__ == "__main__":
    main()
