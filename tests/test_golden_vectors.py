"""Committed golden vectors (tests/golden/vectors_v1.npz, made by tests/golden/make_vectors.py from the
oracle at the time the semantics were pinned): the oracle must still reproduce them (CPU suite) and so must
the HIP path through the C ABI (GPU suite)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V = np.load(os.path.join(ROOT, "tests", "golden", "vectors_v1.npz"))


def _files():
    return [V["file%d_k" % f] for f in range(3)], [V["file%d_t" % f] for f in range(3)]


def _records(h_fn):
    """per-record minimizer through a per-sequence function (oracle API)"""
    cuts, bases = V["seq_cuts"], V["seq_bases"]
    hs, ps = [], []
    for r in range(len(cuts) - 1):
        try:
            h, p = h_fn(bases[int(cuts[r]):int(cuts[r + 1])])
        except ValueError:
            continue
        hs.append(h); ps.append(p)
    return np.concatenate(hs), np.concatenate(ps)


def _check(name, got):
    gk, gt = got
    assert np.array_equal(gk, V[name + "_k"]), name
    assert np.array_equal(gt, V[name + "_t"]), name


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_golden_vectors():
    from oracle import oracle as O
    tax = O.Taxonomy(V["tree_child"], V["tree_parent"])
    files, taxs = _files()
    ok, ot = O.union(files, taxs, tax)
    o = np.argsort(ok, kind="stable")
    _check("union", (ok[o], ot[o]))
    _check("inter", O.inter(files, taxs, tax))
    _check("diff", O.diff(files, taxs, tax))
    _check("diff_t", O.diff(files, taxs, tax, compare_taxid=True))
    _check("common2", O.common(files, 2, taxs, tax))
    _check("merge_u", O.merge_k(files, taxs, mode=O.UNIQUE, tax=tax))
    _check("merge_d", O.merge_k(files, taxs, mode=O.REPEATED, tax=tax))
    _check("merge_d_round1", O.merge_k(files, taxs, mode=O.REPEATED, final_round=False, tax=tax))
    for name, mode in (("uniq", O.UNIQUE), ("rep", O.REPEATED), ("single", O.SINGLETON), ("chunk", O.REPEATED_CHUNK)):
        _check("scan_" + name, O.unique(V["multi_k"], V["multi_t"], mode=mode, tax=tax))
    b, c = V["seq_bases"], V["seq_cuts"]
    assert np.array_equal(O.count_windows(b, c, 31, canonical=True), V["enc_k31_canon"])
    assert np.array_equal(O.count_windows(b, c, 5, canonical=False, circular=True), V["enc_k5_circ"])
    assert np.array_equal(O.count_windows(b, c, 51, hashed=True, canonical=True), V["nt_k51_canon"])
    assert np.array_equal(O.count_windows(b, c, 16, hashed=True, canonical=False, circular=True), V["nt_k16_fwd_circ"])
    assert np.array_equal(O.count_windows(b, c, 21, hashed=True, canonical=True, max_hash=O.max_hash(7)), V["nt_k21_scale7"])
    h, p = _records(lambda s: O.minimizer(s, 21, 9))
    assert np.array_equal(h, V["mini_k21_w9_h"]) and np.array_equal(p, V["mini_k21_w9_p"])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors():
    from unikmer_amd import lib as L
    ctx = L.Context(0)
    ctx.taxonomy_load(V["tree_child"], V["tree_parent"])
    files, taxs = _files()
    _check("union", ctx.union(files, taxs))
    _check("inter", ctx.inter(files, taxs))
    _check("diff", ctx.diff(files, taxs))
    _check("diff_t", ctx.diff(files, taxs, compare_taxid=True))
    _check("common2", ctx.common(files, 2, taxs))
    _check("merge_u", ctx.merge_k(files, taxs, mode=L.UNIQUE))
    _check("merge_d", ctx.merge_k(files, taxs, mode=L.REPEATED))
    _check("merge_d_round1", ctx.merge_k(files, taxs, mode=L.REPEATED, final_round=False))
    for name, mode in (("uniq", L.UNIQUE), ("rep", L.REPEATED), ("single", L.SINGLETON), ("chunk", L.REPEATED_CHUNK)):
        _check("scan_" + name, ctx.unique(V["multi_k"], V["multi_t"], mode=mode))
    # the 2-way kernel directly: (file0 op file1)
    gk, gt = ctx.setop2(L.OP_UNION, files[0], files[1], taxs[0], taxs[1])
    ek, et = ctx.union(files[:2], taxs[:2])
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    b, c = V["seq_bases"], V["seq_cuts"]
    assert np.array_equal(ctx.encode_kmers(b, c, 31, canonical=True), V["enc_k31_canon"])
    assert np.array_equal(ctx.encode_kmers(b, c, 5, canonical=False, circular=True), V["enc_k5_circ"])
    assert np.array_equal(ctx.nthash(b, c, 51, canonical=True), V["nt_k51_canon"])
    assert np.array_equal(ctx.nthash(b, c, 16, canonical=False, circular=True), V["nt_k16_fwd_circ"])
    assert np.array_equal(ctx.nthash(b, c, 21, canonical=True, max_hash=ctx.max_hash(7)), V["nt_k21_scale7"])
    h, p = ctx.minimizer(b, c, 21, 9, with_pos=True)
    assert np.array_equal(h, V["mini_k21_w9_h"]) and np.array_equal(p, V["mini_k21_w9_p"])
