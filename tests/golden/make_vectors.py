#!/usr/bin/env python
"""Generates tests/golden/vectors_v1.npz: small seeded inputs and the outputs the CPU oracle gives for
them (the oracle itself is pinned by the reference's published known answers, tests/test_oracle_kat.py).
The file is data only; both the oracle (CPU suite) and the HIP path (GPU suite) are compared with it, so a
simultaneous drift of oracle and kernels cannot go unnoticed.  Re-run only when semantics change on purpose:
    python tests/golden/make_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_tree  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    rng = np.random.default_rng(20240807)
    v = {}
    child, parent = synth_tree(4, 3)
    v["tree_child"], v["tree_parent"] = np.asarray(child, np.uint32), np.asarray(parent, np.uint32)
    tax = O.Taxonomy(child, parent)
    T = len(child)
    # three sorted sets with taxids (incl. taxid 0), overlapping
    files, taxs = [], []
    for f in range(3):
        k = np.unique(rng.integers(0, 5000, 1800).astype(np.uint64))
        files.append(k)
        taxs.append(rng.integers(0, T + 1, len(k)).astype(np.uint32))
        v["file%d_k" % f], v["file%d_t" % f] = k, taxs[-1]
    for name, (ok, ot) in {
        "union": O.union(files, taxs, tax), "inter": O.inter(files, taxs, tax), "diff": O.diff(files, taxs, tax),
        "diff_t": O.diff(files, taxs, tax, compare_taxid=True), "common2": O.common(files, 2, taxs, tax),
        "merge_u": O.merge_k(files, taxs, mode=O.UNIQUE, tax=tax), "merge_d": O.merge_k(files, taxs, mode=O.REPEATED, tax=tax),
        "merge_d_round1": O.merge_k(files, taxs, mode=O.REPEATED, final_round=False, tax=tax),
    }.items():
        o = np.argsort(ok, kind="stable") if name == "union" else np.arange(len(ok))
        v[name + "_k"], v[name + "_t"] = ok[o], ot[o]
    # a multiset stream and the scan modes
    m = np.sort(rng.integers(0, 700, 3000).astype(np.uint64))
    mt = rng.integers(1, T + 1, len(m)).astype(np.uint32)
    v["multi_k"], v["multi_t"] = m, mt
    for name, mode in (("uniq", O.UNIQUE), ("rep", O.REPEATED), ("single", O.SINGLETON), ("chunk", O.REPEATED_CHUNK)):
        ok, ot = O.unique(m, mt, mode=mode, tax=tax)
        v["scan_%s_k" % name], v["scan_%s_t" % name] = ok, ot
    # sequences: ragged records with IUPAC / lower case, windows of all kinds
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtNRYn", dtype=np.uint8)
    lens = np.array([0, 7, 150, 31, 2500, 3, 64, 1000], dtype=np.uint64)
    cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = alphabet[rng.integers(0, len(alphabet), int(cuts[-1]))]
    v["seq_bases"], v["seq_cuts"] = bases, cuts
    v["enc_k31_canon"] = O.count_windows(bases, cuts, 31, canonical=True)
    v["enc_k5_circ"] = O.count_windows(bases, cuts, 5, canonical=False, circular=True)
    v["nt_k51_canon"] = O.count_windows(bases, cuts, 51, hashed=True, canonical=True)
    v["nt_k16_fwd_circ"] = O.count_windows(bases, cuts, 16, hashed=True, canonical=False, circular=True)
    v["nt_k21_scale7"] = O.count_windows(bases, cuts, 21, hashed=True, canonical=True, max_hash=O.max_hash(7))
    hs, ps = [], []
    for r in range(len(lens)):
        try:
            h, p = O.minimizer(bases[int(cuts[r]):int(cuts[r + 1])], 21, 9)
        except ValueError:
            continue
        hs.append(h); ps.append(p)
    v["mini_k21_w9_h"], v["mini_k21_w9_p"] = np.concatenate(hs), np.concatenate(ps)
    out = os.path.join(HERE, "vectors_v1.npz")
    np.savez_compressed(out, **v)
    print("wrote", out, os.path.getsize(out), "bytes,", len(v), "arrays")


if __name__ == "__main__":
    main()
