#!/usr/bin/env python
"""Per-op device timings (hipEvent, via the C ABI) on device-resident synthetic data.
Usage: python tools/perf_ops.py [--n 1e8] [--ops setop,sort,unique,encode,nthash,tax]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e8)
    ap.add_argument("--ops", default="setop,sort,unique,encode,nthash")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from unikmer_amd import lib
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    n = int(args.n)
    ops = args.ops.split(",")
    res = {}

    def best(fn, reps=args.reps, kernel=False):
        ts = []
        for _ in range(reps):
            fn()
            ts.append(ctx.last_kernel_ms() if kernel else ctx.last_call_ms())
        return min(ts), float(np.median(ts))

    if "setop" in ops or "unique" in ops or "tax" in ops or "pcie" in ops:
        A, B = bench.gen_sets_device((4 * n + 2) // 3, 32 if n > 2e8 else 30, 0, bench.SEED, dev)
        na, nb = A.numel(), B.numel()
    if "setop" in ops:
        out = torch.empty(na + nb, dtype=torch.int64, device=dev)
        for name, op in (("union", lib.OP_UNION), ("inter", lib.OP_INTER), ("diff", lib.OP_DIFF)):
            r = [0]

            def f():
                r[0] = ctx.setop2(op, A, B, out=out).numel()
            f()
            tk, _ = best(f, kernel=True)
            tc, _ = best(f)
            byt = 8 * (na + nb) + 8 * r[0]
            res[name] = {"kernel_ms": tk, "call_ms": tc, "GBps_kernel": byt / tk / 1e6,
                         "kmers_per_s": (na + nb) / tk * 1e3}
    if "pcie" in ops:
        # the boundary handed HOST buffers (what a cgo caller with Go slices does): H2D + kernel + D2H
        Ah, Bh = A.cpu().numpy().view(np.uint64), B.cpu().numpy().view(np.uint64)
        outh = np.empty(na + nb, dtype=np.uint64)
        import time as _t
        ts = []
        for _ in range(3):
            t0 = _t.perf_counter()
            r = ctx.setop2(lib.OP_UNION, Ah, Bh, out=outh)
            ts.append(_t.perf_counter() - t0)
        byt = 8 * (na + nb) + 8 * len(r)
        res["union_host_buffers"] = {"wall_ms": min(ts) * 1e3, "kmers_per_s": (na + nb) / min(ts), "GBps_over_pcie": byt / min(ts) / 1e9,
                                     "note": "pageable numpy buffers in, pageable out"}
    if "tax" in ops:
        from conftest import synth_tree
        child, parent = synth_tree(7, 8)
        ctx.taxonomy_load(child, parent)
        T = len(child)
        ta = (1 + (bench.splitmix64_torch(A ^ 12345) & ((1 << 40) - 1)) % T).to(torch.int32)
        tb = (1 + (bench.splitmix64_torch(B ^ 54321) & ((1 << 40) - 1)) % T).to(torch.int32)
        out = torch.empty(na + nb, dtype=torch.int64, device=dev)
        outt = torch.empty(na + nb, dtype=torch.int32, device=dev)
        for name, op in (("union_tax", lib.OP_UNION), ("inter_tax", lib.OP_INTER)):
            r = [0]

            def f():
                r[0] = ctx.setop2(op, A, B, ta, tb, out=out, out_taxids=outt)[0].numel()
            f()
            tk, _ = best(f, kernel=True)
            byt = 12 * (na + nb) + 12 * r[0]
            res[name] = {"kernel_ms": tk, "GBps_kernel": byt / tk / 1e6, "kmers_per_s": (na + nb) / tk * 1e3}
        # the realistic shapes: (i) each file carries ONE taxid (a genome's k-mers: `count -t`), (ii) taxids
        # clustered by code prefix over 4096 leaves (k-mers of one clade sit together after LCA assignment)
        leaves = T - 8 ** 7 + 1
        shapes = {
            "one_taxid_per_file": (torch.full_like(ta, leaves + 5), torch.full_like(tb, leaves + 77)),
            "clustered_4096_leaves": ((leaves + ((A >> 50) & 4095)).to(torch.int32), (leaves + (((B >> 50) + 1) & 4095)).to(torch.int32)),
        }
        for sname, (xa, xb) in shapes.items():
            for name, op in (("union_tax", lib.OP_UNION), ("inter_tax", lib.OP_INTER)):
                def f():
                    r[0] = ctx.setop2(op, A, B, xa, xb, out=out, out_taxids=outt)[0].numel()
                f()
                tk, _ = best(f, kernel=True)
                byt = 12 * (na + nb) + 12 * r[0]
                res[name + ":" + sname] = {"kernel_ms": tk, "GBps_kernel": byt / tk / 1e6, "kmers_per_s": (na + nb) / tk * 1e3}
        # round 5: the same files with their ONE taxid handed over as a scalar (ukm_setop2_ft); algorithmic bytes: 8 B per
        # input record, 12 per output record
        fa, fb = int(leaves + 5), int(leaves + 77)
        for name, op, fl in (("union_tax", lib.OP_UNION, 0), ("inter_tax", lib.OP_INTER, 0), ("diff_tax", lib.OP_DIFF, 0), ("diff_t_tax", lib.OP_DIFF, lib.F_CMP_TAXID)):
            def f():
                r[0] = ctx.setop2(op, A, B, fa, fb, flags=fl, out=out, out_taxids=outt)[0].numel()
            f()
            tk, _ = best(f, kernel=True)
            tc, _ = best(f)
            byt = 8 * (na + nb) + 12 * r[0]
            res[name + ":file_taxid_scalar"] = {"kernel_ms": tk, "call_ms": tc, "GBps_kernel": byt / tk / 1e6, "kmers_per_s": (na + nb) / tk * 1e3}
        for name, op in (("union_tax", lib.OP_UNION), ("inter_tax", lib.OP_INTER)):
            def f():
                r[0] = ctx.setop2(op, A, B, ta, fb, out=out, out_taxids=outt)[0].numel()
            f()
            tk, _ = best(f, kernel=True)
            byt = 12 * na + 8 * nb + 12 * r[0]
            res[name + ":per_record_x_file_taxid"] = {"kernel_ms": tk, "GBps_kernel": byt / tk / 1e6, "kmers_per_s": (na + nb) / tk * 1e3}
    if "unique" in ops:
        cat = torch.cat([A, B])
        ctx.sort_u64(cat, 62)
        out = torch.empty(cat.numel(), dtype=torch.int64, device=dev)
        r = [0]

        def f():
            r[0] = ctx.unique(cat, out=out).numel()
        f()
        t, _ = best(f)
        res["unique"] = {"call_ms": t, "GBps": (8 * cat.numel() + 8 * r[0]) / t / 1e6, "n": cat.numel()}
        del cat
    if "sort" in ops:
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
        work = torch.empty_like(keys)

        def f():
            work.copy_(keys)
            torch.cuda.synchronize()
            ctx.sort_u64(work, 62)
        f()
        t, med = best(f)
        assert bool((work[1:] >= work[:-1]).all())
        res["sort_u64_62bit"] = {"call_ms": t, "median_ms": med, "keys_per_s": n / t * 1e3,
                                 # bytes the route moves (one histogram read, two scatter passes, one LDS bucket pass): THE figure
                                 "GBps_moved(8n+2*16n+16n)": (8 * n + 2 * 16 * n + 16 * n) / t / 1e6,
                                 "frac_of_8TBps": (8 * n + 2 * 16 * n + 16 * n) / t / 1e6 / 8000.0,
                                 # (SURVEY 8(d)'s 8-pass LSD count, 8n + 8 * 16n, describes a route that no longer exists: footnote only)
                                 "survey_formula_bytes(8n+8*16n)": float(8 * n + 8 * 16 * n)}
        # library reference on the same box: torch.sort -> rocPRIM radix sort (64-bit keys, all 8 bytes)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(4):
            ev0.record()
            sk = torch.sort(keys).values
            ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
            del sk
        res["torch_sort_rocprim_int64"] = {"ms": min(ts), "keys_per_s": n / min(ts) * 1e3}
        vals = torch.arange(n, dtype=torch.int32, device=dev)
        wv = torch.empty_like(vals)

        def f2():
            work.copy_(keys)
            wv.copy_(vals)
            torch.cuda.synchronize()
            ctx.sort_pairs(work, wv, 62)
        f2()
        t, med = best(f2)
        res["sort_pairs_62bit"] = {"call_ms": t, "keys_per_s": n / t * 1e3}
        del keys, work, vals, wv
    if "encode" in ops or "nthash" in ops:
        nb_ = n
        i = torch.arange(nb_, dtype=torch.int64, device=dev)
        w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
        code = (w >> (2 * (i & 31))) & 3
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        bases = lut[code]
        del i, w, code
        nrec = 100
        off = torch.tensor([nb_ * r // nrec for r in range(nrec + 1)], dtype=torch.int64, device=dev)
        out = torch.empty(nb_, dtype=torch.int64, device=dev)
        if "encode" in ops:
            def f():
                ctx.encode_kmers(bases, off, 31, out=out)
            f()
            t, _ = best(f)
            res["encode_k31"] = {"call_ms": t, "bases_per_s": nb_ / t * 1e3, "GBps": 9 * nb_ / t / 1e6}
        if "nthash" in ops:
            def f():
                ctx.nthash(bases, off, 51, out=out)
            f()
            t, _ = best(f)
            res["nthash_k51"] = {"call_ms": t, "bases_per_s": nb_ / t * 1e3}
            mh = ctx.max_hash(1000)

            def f():
                ctx.nthash(bases, off, 51, max_hash=mh, out=out)
            f()
            t, _ = best(f)
            res["nthash_k51_scaled1000"] = {"call_ms": t, "bases_per_s": nb_ / t * 1e3}
            reads = torch.arange(0, nb_ + 1, 150, dtype=torch.int64, device=dev)
            if reads[-1].item() != nb_:
                reads = torch.cat([reads, torch.tensor([nb_], dtype=torch.int64, device=dev)])

            def f():
                ctx.nthash(bases, reads, 51, max_hash=mh, out=out)
            f()
            t, _ = best(f)
            res["nthash_k51_scaled1000_reads150"] = {"call_ms": t, "bases_per_s": nb_ / t * 1e3}
            for kk, ww in ((31, 15), (23, 5), (21, 200)):
                def f():
                    return ctx.minimizer(bases, off, kk, ww)
                f()
                t, _ = best(f)
                res["minimizer_k%d_w%d" % (kk, ww)] = {"call_ms": t, "bases_per_s": nb_ / t * 1e3, "emitted": int(f().numel())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
