#!/usr/bin/env python
"""Measures the BASELINE.json `configs` (other than the headline metric, which is bench.py) on
one MI355X with device-resident synthetic inputs (SURVEY.md §8(d) generators), through the C ABI.

  config 1  union of 2 sorted sets, k=21, 1e6 k-mers each
  config 2  count + sort, k=31, synthetic 100 Mbp FASTA (encode + radix sort + unique)
  config 3  union of 100 sorted files x N k-mers (N = 1e8 in BASELINE; --files3-size scales it),
            one universe of 2N, membership p = 0.5, on ONE GPU
  config 4  inter + diff across 1000 files with taxids, 1e6 k-mers each, p = 0.9, 8-ary depth-7 tree
  config 5  ntHash Scaled-MinHash sketch, k=51, scale=1000, 150 bp reads (--bases5; BASELINE 1e10)

Prints one JSON object; tools are test/measurement helpers, not the product.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,4,5")
    ap.add_argument("--files3", type=int, default=100)
    ap.add_argument("--files3-size", type=float, default=1e8, help="expected k-mers per file (BASELINE: 1e8)")
    ap.add_argument("--files4", type=int, default=1000)
    ap.add_argument("--files4-size", type=float, default=1e6)
    ap.add_argument("--bases5", type=float, default=1e10)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    from unikmer_amd import lib
    from conftest import synth_tree
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    res = {}
    want = set(args.configs.split(","))

    def wall(fn, reps=args.reps):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3, out

    def synth_bases(n):
        i = torch.arange(n, dtype=torch.int64, device=dev)
        w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
        code = (w >> (2 * (i & 31))) & 3
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        return lut[code]

    if "1" in want:
        A, B = bench.gen_sets_device((4 * 1000000 + 2) // 3, 22, 0, bench.SEED, dev)
        out = torch.empty(A.numel() + B.numel(), dtype=torch.int64, device=dev)
        ms, u = wall(lambda: ctx.setop2(lib.OP_UNION, A, B, out=out), reps=10)
        res["config1_union_k21_2x1e6"] = {"ms": ms, "kmers_per_s": (A.numel() + B.numel()) / ms * 1e3, "out": u.numel(),
                                          "note": "launch/latency bound at this size (one partition + one tile kernel + 16-byte readback)"}

    if "2" in want:
        nb = 100_000_000
        bases = synth_bases(nb)
        off = torch.tensor([nb * r // 100 for r in range(101)], dtype=torch.int64, device=dev)
        codes = torch.empty(nb, dtype=torch.int64, device=dev)
        uniq = torch.empty(nb, dtype=torch.int64, device=dev)
        t = {}

        def run():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = ctx.encode_kmers(bases, off, 31, canonical=True, out=codes)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ctx.sort_u64(c, 62)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            u = ctx.unique(c, out=uniq)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            t["encode"], t["sort"], t["unique"] = (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
            return u
        ms, u = wall(run)
        res["config2_count_sort_k31_100Mbp"] = {"ms": ms, "bases_per_s": nb / ms * 1e3, "distinct": u.numel(),
                                                "phases_ms": dict(t), "windows": nb - 100 * 30}
        del bases, codes, uniq

    if "3" in want:
        nfiles, per = args.files3, int(args.files3_size)
        nu = 2 * per
        j = torch.arange(nu, dtype=torch.int64, device=dev)
        gaps = 1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1))
        U = torch.cumsum(gaps, 0)
        del gaps
        files = []
        for f in range(nfiles):
            h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
            files.append(U[(h & 1) == 1])
        del j
        total = sum(x.numel() for x in files)
        out = torch.empty(min(total, nu) + 8, dtype=torch.int64, device=dev)
        ctx.union(files, out=out)  # untimed: the first call grows the workspace and first-touches ~20 GB of it (1.5-3 s)
        ms, u = wall(lambda: ctx.union(files, out=out), reps=args.reps)
        assert u.numel() <= nu and bool((u[1:] > u[:-1]).all())
        n_probe, x_probe = u.numel(), int(bench._xor_fold(u)) if hasattr(bench, "_xor_fold") else 0
        os.environ["UKM_PUNION"] = "0"   # the k-way streaming merge alone (what answers when the sets do not overlap)
        ctx.union(files, out=out)
        ms_kway, uk = wall(lambda: ctx.union(files, out=out), reps=args.reps)
        del os.environ["UKM_PUNION"]
        assert uk.numel() == n_probe and (not hasattr(bench, "_xor_fold") or int(bench._xor_fold(uk)) == x_probe)
        res["config3_union_%d_files_x_%.0e" % (nfiles, per)] = {"ms": ms, "input_kmers": total, "kmers_per_s": total / ms * 1e3,
                                                                  "out": u.numel(), "gpus": 1,
                                                                  "algorithmic_GB": (8 * total + 8 * u.numel()) / 1e9,
                                                                  "ms_kway_merge_only": ms_kway,
                                                                  "note": "union by LDS hash probes against the union of the first eight files (ukm_punion.hip); "
                                                                          "ms_kway_merge_only = the k-way streaming merge (ukm_kway.hip, UKM_PUNION=0), same "
                                                                          "result (size and XOR checksum compared); round 1: 7-level pairwise tree, 78.7 ms"}
        del files, U, out

    if "4" in want:
        nfiles, per = args.files4, int(args.files4_size)
        child, parent = synth_tree(7, 8)
        ctx.taxonomy_load(child, parent)
        T = len(child)
        nu = int(per / 0.9)
        j = torch.arange(nu, dtype=torch.int64, device=dev)
        gaps = 1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1))
        U = torch.cumsum(gaps, 0)
        files, taxs = [], []
        thr = int(0.9 * (1 << 20))
        for f in range(nfiles):
            h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
            k = U[((h >> 11) & ((1 << 20) - 1)) < thr]
            files.append(k)
            taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
        total = sum(x.numel() for x in files)
        ok = torch.empty(files[0].numel() + 8, dtype=torch.int64, device=dev)
        ot = torch.empty(files[0].numel() + 8, dtype=torch.int32, device=dev)
        ms_i, ri = wall(lambda: ctx.inter(files, taxs, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        n_inter = ri[0].numel()
        ms_d, rd = wall(lambda: ctx.diff(files, taxs, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        ms_dt, rdt = wall(lambda: ctx.diff(files, taxs, compare_taxid=True, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        res["config4_inter_diff_%d_files_taxids" % nfiles] = {
            "inter_ms": ms_i, "inter_out": n_inter, "diff_ms": ms_d, "diff_out": rd[0].numel(),
            "diff_compare_taxid_ms": ms_dt, "diff_compare_taxid_out": rdt[0].numel(), "input_kmers": total,
            "inter_kmers_per_s": total / ms_i * 1e3,
            "note": "SURVEY 8(d) generator (independent p = 0.9 draws): 0.9^n empties the running result after ~130 files, so "
                    "inter and plain diff are EARLY-EXIT dominated (inter.go:283-286, diff.go:457-459); only diff -t visits all "
                    "files.  The `_core` entry below is the same job with a result that survives every file."}
        # the same shape with a shared core (30 % of the universe is in every file) and private codes in the first
        # file: inter, diff and diff -t all fold over ALL files
        core = ((bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 77)) >> 11) & ((1 << 20) - 1)) < int(0.3 * (1 << 20))
        files2, taxs2 = [], []
        for f in range(nfiles):
            h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
            m = (((h >> 11) & ((1 << 20) - 1)) < thr) | core
            if f == 0:
                k = torch.cat([U[m], U[-1] + 1 + torch.arange(per // 10, dtype=torch.int64, device=dev) * 3])
            else:
                k = U[m]
            files2.append(k)
            taxs2.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
        total2 = sum(x.numel() for x in files2)
        ok = torch.empty(files2[0].numel() + 8, dtype=torch.int64, device=dev)
        ot = torch.empty(files2[0].numel() + 8, dtype=torch.int32, device=dev)
        ms_i, ri = wall(lambda: ctx.inter(files2, taxs2, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        n_inter = ri[0].numel()
        ms_d, rd = wall(lambda: ctx.diff(files2, taxs2, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        n_diff = rd[0].numel()
        ms_dt, rdt = wall(lambda: ctx.diff(files2, taxs2, compare_taxid=True, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        # `common` of the same files with the default threshold (every file): the probe fold again; the counting merge
        # (k-way keep-all merge + threshold scan) timed beside it
        okc = torch.empty(total2 + 8, dtype=torch.int64, device=dev)
        otc = torch.empty(total2 + 8, dtype=torch.int32, device=dev)
        ms_c, rc = wall(lambda: ctx.common(files2, nfiles, taxs2, out=okc, out_taxids=otc), reps=max(1, args.reps - 1))
        assert rc[0].numel() == n_inter
        sums = (int(rc[0].sum()), int(rc[1].long().sum()))
        os.environ["UKM_COMMON_PROBE"] = "0"
        ms_cm, rcm = wall(lambda: ctx.common(files2, nfiles, taxs2, out=okc, out_taxids=otc), reps=1)
        del os.environ["UKM_COMMON_PROBE"]
        assert rcm[0].numel() == n_inter and sums == (int(rcm[0].sum()), int(rcm[1].long().sum()))
        del okc, otc
        res["config4_core_inter_diff_%d_files_taxids" % nfiles] = {
            "common_all_files_ms": ms_c, "common_all_files_counting_merge_ms": ms_cm,
            "inter_ms": ms_i, "inter_out": n_inter, "diff_ms": ms_d, "diff_out": n_diff,
            "diff_compare_taxid_ms": ms_dt, "diff_compare_taxid_out": rdt[0].numel(), "input_kmers": total2,
            "inter_kmers_per_s": total2 / ms_i * 1e3, "diff_kmers_per_s": total2 / ms_d * 1e3,
            "note": "no early exit: every one of the %d links runs (result sizes above are non-zero)" % (nfiles - 1)}
        del files2, taxs2
        del files, taxs, U

    if "5" in want:
        nb = int(args.bases5)
        nb -= nb % 150
        chunk = 1 << 31  # generate in pieces to bound torch temporaries
        parts = []
        for lo in range(0, nb, chunk):
            hi = min(lo + chunk, nb)
            i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
            w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
            code = ((w >> (2 * (i & 31))) & 3).to(torch.uint8)
            del i, w
            parts.append(code)
        code = torch.cat(parts) if len(parts) > 1 else parts[0]
        del parts
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        bases = lut[code.long()] if nb <= (1 << 31) else torch.cat([lut[code[a:a + chunk].long()] for a in range(0, nb, chunk)])
        del code
        reads = torch.arange(0, nb + 1, 150, dtype=torch.int64, device=dev)
        mh = ctx.max_hash(1000)
        cap = nb // 500 + 1024
        out = torch.empty(cap, dtype=torch.int64, device=dev)
        uq = torch.empty(cap, dtype=torch.int64, device=dev)
        t = {}

        def run5():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h = ctx.nthash(bases, reads, 51, canonical=True, max_hash=mh, out=out)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ctx.sort_u64(h, int(mh).bit_length())  # kept hashes are <= maxHash: 55 significant bits at scale 1000
            u = ctx.unique(h, out=uq)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            t["nthash+filter"], t["sort+unique"] = (t1 - t0) * 1e3, (t2 - t1) * 1e3
            return h.numel(), u.numel()
        ms, (kept, distinct) = wall(run5, reps=max(1, args.reps - 1))
        res["config5_nthash_scaled1000_k51_reads150"] = {"ms": ms, "bases": nb, "bases_per_s": nb / ms * 1e3, "kept": kept,
                                                          "distinct": distinct, "phases_ms": dict(t),
                                                          "GBps_read": nb / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
