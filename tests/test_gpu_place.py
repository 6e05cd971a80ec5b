"""The keep-everything merge of many files that share their codes, by PLACEMENT (csrc/ukm_punion.hip, pl_merge_kernel;
`merge` = mergeChunksFile's heap, util-sort.go:196-225,289-351): the distinct codes from the probe union, a count per code
from one probe pass, every code's run written in one piece, the TaxIds placed eight files at a time -- against a stable
sort of the concatenation (= the heap's order: equal codes in file order) and the oracle's modes.

UKM_PLACE=1 / UKM_PUNION=1 force the routes at test sizes (the library takes them from 96 files and 2^26 records on when a
workgroup's slice of a file is long enough); `ctx.last_route() == 7` shows the placement answered."""
import numpy as np
import pytest

from conftest import splitmix64, synth_tree

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572
ROUTE_PLACE = 7


@pytest.fixture(scope="module")
def env():
    from oracle import oracle as O
    from unikmer_amd import lib as L
    ctx = L.Context(0)
    child, parent = synth_tree(5, 8)
    ctx.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    yield O, L, ctx, tax, len(child)
    ctx.close()


def _universe(n, gap_bits=24, seed=SEED):
    j = np.arange(n, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64(np.uint64(seed) ^ j) & np.uint64((1 << gap_bits) - 1))
    return np.cumsum(gaps, dtype=np.uint64)


def _member(n, f, p, seed):
    h = splitmix64(np.uint64(seed + 1000 * (f + 1)) ^ np.arange(n, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p


def _taxids(codes, T, salt):
    return (np.uint64(1) + splitmix64(np.uint64(SEED + 2 + salt) ^ codes) % np.uint64(T)).astype(np.uint32)


def _stable(streams, taxs=None):
    cat = np.concatenate(streams)
    o = np.argsort(cat, kind="stable")
    return (cat[o], np.concatenate(taxs)[o]) if taxs is not None else cat[o]


@pytest.mark.parametrize("n_univ,nfiles,p", [(3000, 40, 0.7), (20000, 100, 0.5), (1500, 300, 0.8), (50000, 33, 0.6), (700, 17, 0.9),
                                              (2000, 1500, 0.6)])
def test_placement_merge_equals_the_stable_sort(env, monkeypatch, n_univ, nfiles, p):
    """plain and with taxids; one range and many; 17 and 33 files (one file past a batch); 1500 files (more than
    the single-pass merge takes); the -u / -d scans behind the merged sequence"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_PLACE", "1")
    monkeypatch.setenv("UKM_PUNION", "1")
    U = _universe(n_univ)
    files = [U[_member(len(U), f, p, 7)] for f in range(nfiles)]
    files = [f for f in files if len(f)]
    taxs = [_taxids(f + np.uint64(i), T, i) for i, f in enumerate(files)]
    ek, et = _stable(files, taxs)
    gk, gt = ctx.merge_k(files, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_PLACE
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    assert np.array_equal(ctx.merge_k(files, mode=L.PLAIN), ek)
    assert ctx.last_route() == ROUTE_PLACE
    some = [t if i % 3 else None for i, t in enumerate(taxs)]          # files without taxids among files with
    gk, gt = ctx.merge_k(files, some, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_PLACE
    ok, ot = O.merge_k(files, some, mode=O.PLAIN, tax=tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    monkeypatch.setenv("UKM_PUNION", "0")                               # (the scans, not the counting probes, behind the merge)
    for final in (True, False):
        gk, gt = ctx.merge_k(files, taxs, mode=L.REPEATED, final_round=final)
        ok, ot = O.merge_k(files, taxs, mode=O.REPEATED, final_round=final, tax=tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), final


def test_placement_declines_what_it_cannot_place(env, monkeypatch):
    """a code twice inside one file (two records for one cell), an unsorted file, all-ones codes (the tables' empty marker),
    files that share nothing: the other merges answer, same result; one-record files and a file that ends inside every
    range are fine"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_PLACE", "1")
    monkeypatch.setenv("UKM_PUNION", "1")
    rng = np.random.default_rng(3)
    U = _universe(6000)
    files = [U[_member(len(U), f, 0.6, 9)] for f in range(40)]
    files[5] = files[5][:1]
    files[6] = files[6][-1:]
    files[7] = files[7][: len(files[7]) // 3]
    taxs = [_taxids(f + np.uint64(i), T, i) for i, f in enumerate(files)]
    ek, et = _stable(files, taxs)
    gk, gt = ctx.merge_k(files, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_PLACE
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    cases = {}
    d = list(files); d[11] = np.sort(np.concatenate([d[11], d[11][::9]])); cases["duplicate"] = d
    d = list(files); d[12] = rng.permutation(d[12]); cases["unsorted"] = d
    d = list(files); d[13] = np.concatenate([d[13], np.full(1, np.uint64(2**64 - 1))]); d[2] = np.concatenate([d[2], np.full(1, np.uint64(2**64 - 1))])
    cases["all ones"] = d
    cases["disjoint"] = [np.sort(rng.choice(1 << 40, 3000, replace=False).astype(np.uint64) + np.uint64(f << 44)) for f in range(30)]
    for name, fs in cases.items():
        ts = [_taxids(f + np.uint64(i), T, i) for i, f in enumerate(fs)]
        gk, gt = ctx.merge_k(fs, ts, mode=L.PLAIN)
        assert ctx.last_route() != ROUTE_PLACE, name
        if name == "unsorted":   # (the library sorts what is not sorted: a stable sort of the concatenation; taxids here depend on the code only)
            ok, ot = _stable(fs, ts)
            assert np.array_equal(gk, ok) and np.array_equal(np.sort(gt), np.sort(ot)), name
        else:
            ok, ot = O.merge_k(fs, ts, mode=O.PLAIN, tax=tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), name
