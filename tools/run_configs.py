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

    HBM_PEAK = 8.0e12          # B/s (MI355X_MICROARCH.md)
    VALU_PEAK = 256 * 4 * 16 * 2.4e9   # lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes at 2.4 GHz

    def roof_hbm(nbytes, ms, what):
        """the roofline object of a config: ALGORITHMIC bytes (SURVEY 8(d) formulas) over the wall time of the call(s)"""
        a = nbytes / (ms * 1e-3)
        return {"bound": "hbm", "algorithmic_bytes": float(nbytes), "ms": ms, "achieved": a / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": a / HBM_PEAK, "bytes_are": what}

    def roof_valu(lane_instr, ms, what):
        a = lane_instr / (ms * 1e-3)
        return {"bound": "valu", "lane_instructions": float(lane_instr), "ms": ms, "achieved": a / 1e12, "peak": VALU_PEAK / 1e12,
                "unit": "T lane-instr/s", "frac": a / VALU_PEAK, "instructions_are": what}

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
                                          "roofline": roof_hbm(8 * (A.numel() + B.numel()) + 8 * u.numel(), ms, "8(|A|+|B|) read + 8|out| written"),
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
        # the same job through the one-call entry point (round 6: ukm_count = windows -> sort -> unique on the device, one
        # stream synchronisation and one read-back instead of three)
        ms_one, u1 = wall(lambda: ctx.count(bases, off, 31, canonical=True, out=uniq))
        assert u1.numel() == u.numel()
        w = nb - 100 * 30
        res["config2_count_sort_k31_100Mbp"] = {"ms": ms, "bases_per_s": nb / ms * 1e3, "distinct": u.numel(),
                                                "phases_ms": dict(t), "windows": w, "ms_ukm_count_one_call": ms_one,
                                                # THE fraction = bytes the route moves (round-5 review: the survey's 8-pass LSD byte
                                                # count describes a route that no longer exists and gave "fractions" above 1)
                                                "roofline": roof_hbm((nb + 8 * w) + (8 * w + 2 * 16 * w + 16 * w) + (8 * w + 8 * u.numel()), ms,
                                                                     "bytes the route moves: encode 1 B/base + 8 B/window; sort = one histogram read, "
                                                                     "TWO scatter passes and one LDS bucket pass (5.6 GB per 1e8 keys); unique 8n + 8u"),
                                                "roofline_sort": roof_hbm(8 * w + 2 * 16 * w + 16 * w, t["sort"], "sort alone, bytes the route moves"),
                                                "survey_formula": {"bytes": float((nb + 8 * w) + (8 * w + 16 * w * 8) + (8 * w + 8 * u.numel())),
                                                                   "is": "SURVEY 8(d)'s count for an 8-pass LSD sort (8n + 16nP, P = 8): a footnote, "
                                                                         "NOT a roofline fraction -- this route makes three passes over HBM"}}
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
        ctx.set_option("punion", 0)  # the k-way streaming merge alone (what answers when the sets do not overlap)
        ctx.union(files, out=out)
        ms_kway, uk = wall(lambda: ctx.union(files, out=out), reps=args.reps)
        ctx.set_option("punion", None)
        assert uk.numel() == n_probe and (not hasattr(bench, "_xor_fold") or int(bench._xor_fold(uk)) == x_probe)
        res["config3_union_%d_files_x_%.0e" % (nfiles, per)] = {"ms": ms, "input_kmers": total, "kmers_per_s": total / ms * 1e3,
                                                                  "out": u.numel(), "gpus": 1,
                                                                  "algorithmic_GB": (8 * total + 8 * u.numel()) / 1e9,
                                                                  "roofline": roof_hbm(8 * total + 8 * u.numel(), ms, "8 B per input record read + 8 B per output record written"),
                                                                  "roofline_kway_merge_only": roof_hbm(8 * total + 8 * u.numel(), ms_kway, "the same bytes over the k-way merge's time"),
                                                                  "ms_kway_merge_only": ms_kway,
                                                                  "note": "union by LDS hash probes against the union of the first eight files (ukm_punion.hip); "
                                                                          "ms_kway_merge_only = the k-way streaming merge (ukm_kway.hip, UKM_PUNION=0), same "
                                                                          "result (size and XOR checksum compared); round 1: 7-level pairwise tree, 78.7 ms"}
        del uk, u
        ctx.trim()   # (the k-way merge's 160 GB of workspace)
        torch.cuda.empty_cache()
        # the same files WITH taxids (union.go:195-201: LCA over every record of a code): (i) every record of a file carries
        # that file's taxid (k-mers of one genome, `count -t`), (ii) uniformly random taxids
        child, parent = synth_tree(7, 8)
        ctx.taxonomy_load(child, parent)
        T = len(child)
        leaves0 = T - 8 ** 7 + 1
        outt = torch.empty(out.numel(), dtype=torch.int32, device=dev)
        entry = res["config3_union_%d_files_x_%.0e" % (nfiles, per)]
        for kind in ("one_taxid_per_file_scalar", "one_taxid_per_file", "random_taxids"):
            if kind == "one_taxid_per_file_scalar":
                # round 5: the file's taxid handed over as ONE number (ukm_union_ft), nothing expanded
                taxs = [int(leaves0 + (f * 7919) % (8 ** 7)) for f in range(len(files))]
            elif kind == "one_taxid_per_file":
                taxs = [torch.full((x.numel(),), leaves0 + (f * 7919) % (8 ** 7), dtype=torch.int32, device=dev) for f, x in enumerate(files)]
            else:
                taxs = [(1 + (bench.splitmix64_torch(x ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32)
                        for f, x in enumerate(files)]
            ctx.union(files, taxs, out=out, out_taxids=outt)
            ms_t, ut = wall(lambda: ctx.union(files, taxs, out=out, out_taxids=outt), reps=args.reps)
            assert ut[0].numel() == n_probe
            bpr = 8 if kind.endswith("scalar") else 12
            entry["with_" + kind] = {"ms": ms_t, "route": ctx.last_route(), "kmers_per_s": total / ms_t * 1e3,
                                     "checksum": int(ut[0].sum().item()) ^ int(ut[1].to(torch.int64).sum().item()),
                                     "roofline": roof_hbm(bpr * total + 12 * n_probe, ms_t, "%d B per input record read + 12 B per output record written" % bpr)}
            del taxs, ut
        entry["note_taxids"] = ("route 3 = the hash-probe pass with the TaxId fold in its LDS tables (ukm_punion.hip, round 4); through the k-way "
                                "merge the half-size shape took 106 / 124 ms against 31.5 / 87 ms (tools/c3_tax_bench.py)")
        del files, U, out, outt

    if "4" in want:
        ctx.close()
        torch.cuda.empty_cache()
        ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
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
        # the pointer / length tables of the 1000 streams are built once (what a host that keeps its decoded files on the
        # device holds anyway); passing the Python lists instead costs the binding ~0.4 - 0.8 ms per call
        tab2 = ctx.stream_table(files2, taxs2)
        ms_i, ri = wall(lambda: ctx.inter(tab2, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        n_inter = ri[0].numel()
        ms_d, rd = wall(lambda: ctx.diff(tab2, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        n_diff = rd[0].numel()
        ms_dt, rdt = wall(lambda: ctx.diff(tab2, compare_taxid=True, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        ms_i_list, _ = wall(lambda: ctx.inter(files2, taxs2, out=ok, out_taxids=ot), reps=max(1, args.reps - 1))
        # `common` of the same files with the default threshold (every file): the probe fold again; the counting merge
        # (k-way keep-all merge + threshold scan) timed beside it
        okc = torch.empty(total2 + 8, dtype=torch.int64, device=dev)
        otc = torch.empty(total2 + 8, dtype=torch.int32, device=dev)
        ms_c, rc = wall(lambda: ctx.common(files2, nfiles, taxs2, out=okc, out_taxids=otc), reps=max(1, args.reps - 1))
        assert rc[0].numel() == n_inter
        sums = (int(rc[0].sum()), int(rc[1].long().sum()))
        ctx.set_option("common_probe", 0)
        ms_cm, rcm = wall(lambda: ctx.common(files2, nfiles, taxs2, out=okc, out_taxids=otc), reps=1)
        ctx.set_option("common_probe", None)
        assert rcm[0].numel() == n_inter and sums == (int(rcm[0].sum()), int(rcm[1].long().sum()))
        # the same 1000 files through the keep-everything merge (mergeChunksFile), `common` one below the full threshold
        # (counting merge + run scan) and `union` with the taxid fold: the many-stream routes (ukm_srmerge.hip / ukm_kway.hip)
        ms_m, rm = wall(lambda: ctx.merge_k(files2, taxs2, out=okc, out_taxids=otc), reps=max(1, args.reps - 1))
        route_m = ctx.last_route()
        ms_c1, rc1 = wall(lambda: ctx.common(files2, nfiles - 1, taxs2, out=okc, out_taxids=otc), reps=max(1, args.reps - 1))
        ms_u, ru = wall(lambda: ctx.union(files2, taxs2, out=okc, out_taxids=otc), reps=max(1, args.reps - 1))
        route_u = ctx.last_route()
        many = {"merge_ms": ms_m, "merge_route": route_m, "common_threshold_minus_1_ms": ms_c1, "common_threshold_minus_1_out": rc1[0].numel(),
                "union_ms": ms_u, "union_out": ru[0].numel(), "union_route": route_u, "input_kmers": total2,
                "roofline_merge": roof_hbm(24 * total2, ms_m, "12 B per record read + 12 B per record written"),
                "roofline_common_threshold_minus_1": roof_hbm(12 * total2 + 12 * rc1[0].numel(), ms_c1, "12 B per input record + 12 B per output record"),
                "roofline_union": roof_hbm(12 * total2 + 12 * ru[0].numel(), ms_u, "12 B per input record + 12 B per output record"),
                "note": "route 4 = single-pass range merge (ukm_srmerge.hip), 2 = multi-level k-way merge (ukm_kway.hip)"}
        # round 5: the same files with ONE taxid per file handed over as a scalar (ukm_*_ft) beside the plain operations:
        # inter / diff / diff -t / common of all files are the plain operation and a fill
        ftax = [int(1 + (f * 7919) % T) for f in range(nfiles)]
        tabf = ctx.stream_table(files2, ftax)
        tabp = ctx.stream_table(files2)
        per_file = {}
        for name, fn_t, fn_p in (
                ("inter", lambda: ctx.inter(tabf, out=ok, out_taxids=ot), lambda: ctx.inter(tabp, out=ok)),
                ("diff", lambda: ctx.diff(tabf, out=ok, out_taxids=ot), lambda: ctx.diff(tabp, out=ok)),
                ("diff_compare_taxid", lambda: ctx.diff(tabf, compare_taxid=True, out=ok, out_taxids=ot), lambda: ctx.diff(tabp, out=ok)),
                ("common_all_files", lambda: ctx.common(tabf, nfiles, out=okc, out_taxids=otc), lambda: ctx.common(tabp, nfiles, out=okc)),
                ("common_threshold_minus_1", lambda: ctx.common(tabf, nfiles - 1, out=okc, out_taxids=otc), lambda: ctx.common(tabp, nfiles - 1, out=okc)),
                ("union", lambda: ctx.union(tabf, out=okc, out_taxids=otc), lambda: ctx.union(tabp, out=okc)),
                ("merge", lambda: ctx.merge_k(tabf, out=okc, out_taxids=otc), lambda: ctx.merge_k(tabp, out=okc))):
            fn_t(); fn_p()
            mt, rt = wall(fn_t, reps=max(1, args.reps - 1))
            rte = ctx.last_route()
            mp, rp = wall(fn_p, reps=max(1, args.reps - 1))
            per_file[name] = {"ms": mt, "plain_ms": mp, "ratio": mt / mp, "out": rt[0].numel(), "route": rte}
        many["one_taxid_per_file_scalar"] = per_file
        del okc, otc
        res["config4_core_files_merge_common_union"] = many
        res["config4_core_inter_diff_%d_files_taxids" % nfiles] = {
            "common_all_files_ms": ms_c, "common_all_files_counting_merge_ms": ms_cm,
            "inter_ms": ms_i, "inter_ms_python_lists": ms_i_list, "inter_out": n_inter, "diff_ms": ms_d, "diff_out": n_diff,
            "diff_compare_taxid_ms": ms_dt, "diff_compare_taxid_out": rdt[0].numel(), "input_kmers": total2,
            "inter_kmers_per_s": total2 / ms_i * 1e3, "diff_kmers_per_s": total2 / ms_d * 1e3,
            "roofline_inter": roof_hbm(12 * total2 + 12 * n_inter, ms_i, "12 B (code + taxid) per input record read + 12 B per output record"),
            "roofline_diff": roof_hbm(12 * total2 + 12 * n_diff, ms_d, "12 B per input record read + 12 B per output record"),
            "roofline_diff_compare_taxid": roof_hbm(12 * total2 + 12 * rdt[0].numel(), ms_dt, "12 B per input record read + 12 B per output record"),
            "roofline_common_all_files": roof_hbm(12 * total2 + 12 * n_inter, ms_c, "12 B per input record read + 12 B per output record"),
            "note": "no early exit: every one of the %d links runs (result sizes above are non-zero)" % (nfiles - 1)}
        del files2, taxs2, tab2
        del files, taxs, U

    if "5" in want:
        # (a fresh context: the workspace of config 3's k-way merge -- two buffers of 1e10 records -- stays with a context
        #  until it is destroyed, and config 5 needs 30 GB of its own)
        ctx.close()
        torch.cuda.empty_cache()
        ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
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
        L5 = 512   # strip length the kernel picks at this size and scale: a strip of L windows rolls L + k - 1 bases
        steps = nb * (L5 + 50) / L5
        res["config5_nthash_scaled1000_k51_reads150"] = {"ms": ms, "bases": nb, "bases_per_s": nb / ms * 1e3, "kept": kept,
                                                          "distinct": distinct, "phases_ms": dict(t),
                                                          "GBps_read": nb / ms / 1e6,
                                                          "roofline": roof_valu(steps * 15 * 1.0, t["nthash+filter"],
                                                                                "the strip kernel is VALU bound by design: 15 vector instructions per "
                                                                                "rolled base (13 VALU + 2 LDS issue slots; DESIGN 4.4), one lane per strip"),
                                                          "roofline_hbm": roof_hbm(nb + 8 * kept, t["nthash+filter"], "1 B per base read + 8 B per kept hash (for reference: "
                                                                                   "the kernel is not HBM bound)")}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
