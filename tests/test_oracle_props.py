"""CPU-only property tests of the oracle itself (set algebra identities, the README
equivalence `union -s == sort -u == merge -u`, LCA against a brute-force ancestor walk)."""
import numpy as np
import pytest

from conftest import splitmix64, synth_tree
from oracle import oracle as O


def _sets(n, seed=1):
    rng = np.random.default_rng(seed)
    U = np.cumsum(rng.integers(1, 1 << 20, n).astype(np.uint64))
    m = rng.integers(0, 4, n)
    return U[(m == 0) | (m >= 2)], U[(m == 1) | (m >= 2)]


@pytest.fixture(scope="module")
def tree():
    child, parent = synth_tree(depth=4, arity=5)
    return O.Taxonomy(child, parent), dict(zip(child.tolist(), parent.tolist())), len(child)


def _brute_lca(par, a, b):
    if a == 0 or b == 0:
        return 0
    if a == b:
        return a
    if a not in par or b not in par:
        return 0
    anc = set()
    x = a
    while True:
        anc.add(x)
        if par[x] == x:
            break
        x = par[x]
    x = b
    while x not in anc:
        if par[x] == x:
            return 0
        x = par[x]
    return x


def test_lca_bruteforce(tree):
    tax, par, T = tree
    rng = np.random.default_rng(0)
    for a, b in rng.integers(0, T + 5, (3000, 2)):
        assert tax.lca(a, b) == _brute_lca(par, int(a), int(b))
    # associativity / commutativity on valid ids (what makes the GPU's fold order irrelevant)
    for a, b, c in rng.integers(1, T + 1, (500, 3)):
        assert tax.lca(a, b) == tax.lca(b, a)
        assert tax.lca(tax.lca(a, b), c) == tax.lca(a, tax.lca(b, c))


def test_set_algebra_vs_numpy():
    A, B = _sets(50000)
    assert np.array_equal(O.union([A, B]), np.union1d(A, B))
    assert np.array_equal(O.inter([A, B]), np.intersect1d(A, B))
    assert np.array_equal(O.diff([A, B]), np.setdiff1d(A, B))
    assert np.array_equal(O.common([A, B], 2), np.intersect1d(A, B))
    assert np.array_equal(O.common([A, B], 1), np.union1d(A, B))
    assert O.common_threshold(10, 0.75) == 7 and O.common_threshold(3, 1.0) == 3


def test_union_equals_sort_unique_equals_merge(tree):
    # README.md:215-229 (C-9): union -s == sort -u == split + merge -u on the (code, taxid) stream
    tax, par, T = tree
    A, B = _sets(20000, seed=3)
    C = A[::3].copy()
    files = [A, B, C]
    taxs = [(np.uint64(1) + splitmix64(f + np.uint64(i)) % np.uint64(T)).astype(np.uint32) for i, f in enumerate(files)]
    uk, ut = O.union(files, taxs, tax)
    ck, ct = O.sort_pairs(np.concatenate(files), np.concatenate(taxs))
    sk, st = O.unique(ck, ct, mode=O.UNIQUE, tax=tax)
    mk, mt = O.merge_k(files, taxs, mode=O.UNIQUE, tax=tax)
    assert np.array_equal(uk, sk) and np.array_equal(ut, st)
    assert np.array_equal(uk, mk) and np.array_equal(ut, mt)
    # brute-force check of the LCA fold for a few codes
    d = {}
    for f, t in zip(files, taxs):
        for code, tx in zip(f[:200].tolist(), t[:200].tolist()):
            d[code] = _brute_lca(par, d[code], tx) if code in d else tx
    lut = dict(zip(uk.tolist(), ut.tolist()))
    full = {}
    for f, t in zip(files, taxs):
        for code, tx in zip(f.tolist(), t.tolist()):
            full[code] = _brute_lca(par, full[code], tx) if code in full else tx
    assert all(lut[c] == full[c] for c in list(d)[:300])


def test_repeated_protocol_two_rounds():
    # sort -d via chunks: chunk stage (one/two copies) + final merge == in-RAM sort -d
    rng = np.random.default_rng(9)
    x = rng.integers(0, 4000, 20000).astype(np.uint64)
    chunks = [O.sort_u64(x[i::4]) for i in range(4)]
    dumped = [O.unique(c, mode=O.REPEATED_CHUNK) for c in chunks]
    round1 = [O.merge_k(dumped[:2], mode=O.REPEATED, final_round=False),
              O.merge_k(dumped[2:], mode=O.REPEATED, final_round=False)]
    final = O.merge_k(round1, mode=O.REPEATED, final_round=True)
    assert np.array_equal(final, O.unique(O.sort_u64(x), mode=O.REPEATED))
    vals, cnt = np.unique(x, return_counts=True)
    assert np.array_equal(final, vals[cnt > 1])
    # count -u / count -d partition the distinct set (count.go:424-432)
    assert np.array_equal(O.unique(O.sort_u64(x), mode=O.SINGLETON), vals[cnt == 1])


def test_inter_diff_multiset_and_quirks():
    a = np.array([1, 1, 2, 5, 5, 5, 9], dtype=np.uint64)
    b = np.array([1, 5, 5, 7, 9, 9], dtype=np.uint64)
    assert O.inter([a, b]).tolist() == [1, 5, 5, 9]          # equality advances both cursors
    assert O.diff([a, b]).tolist() == [1, 2, 5]              # survivors, duplicates collapsed
    e = np.empty(0, np.uint64)
    assert O.inter([a, e, b]).tolist() == a.tolist()         # inter.go:211-217 quirk
    assert O.inter([e, a]).tolist() == []
    assert O.diff([a, e]).tolist() == [1, 2, 5, 9]


def test_allcores_sorted_merge_matches_reference_algorithms():
    """bench.py's second CPU bar gives the same sets as the restated
    reference loops on strictly increasing inputs, including empty and one-element inputs."""
    rng = np.random.default_rng(3)
    for na, nb in ((0, 0), (0, 7), (5, 0), (1, 1), (1000, 3), (50_000, 60_000)):
        a = np.unique(rng.integers(0, 200_000, na).astype(np.uint64))
        b = np.unique(rng.integers(0, 200_000, nb).astype(np.uint64))
        for op, ref in ((0, O.union), (1, O.inter), (2, O.diff)):
            if op == 1 and len(b) == 0 and len(a):
                continue  # the reference's inter keeps the first set when a later file is empty (inter.go quirk)
            _, got, threads = O.time_setop2_allcores(op, a, b)
            exp = ref([a, b])
            assert threads >= 1
            assert np.array_equal(got, np.sort(exp))


def test_oracle_by_value_ranges_equals_the_oracle_on_whole_files(tree):
    """conftest.oracle_by_value_ranges (how the full-size GPU tests afford the oracle over 1e9 records: value range by value
    range on the host's cores) returns exactly what the oracle returns on the whole files -- union / inter / diff / diff -t /
    common / keep-everything merge with taxids, files that lack records in some ranges, a file that is empty altogether
    (inter.go:211-217: the running result is kept), and a file with a duplicated code (the multiset rule)."""
    from conftest import oracle_by_value_ranges as by_ranges
    tax, _, T = tree
    rng = np.random.default_rng(11)
    U = np.cumsum(rng.integers(1, 1 << 18, 60_000).astype(np.uint64))
    files = [U[rng.random(len(U)) < p] for p in (0.9, 0.8, 0.85, 0.7, 0.95)]
    files[3] = files[3][files[3] > U[len(U) // 3]]                 # nothing in the lowest ranges
    files[1] = np.sort(np.concatenate([files[1], files[1][100:101]]))  # a code twice in one file
    taxs = [rng.integers(1, T + 1, len(f)).astype(np.uint32) for f in files]
    ops = {
        "union": (lambda k, t: O.union(k, t, tax), "set"),
        "inter": (lambda k, t: O.inter(k, t, tax), "inter"),
        "diff": (lambda k, t: O.diff(k, t, tax), "set"),
        "diff_t": (lambda k, t: O.diff(k, t, tax, compare_taxid=True), "set"),
        "common": (lambda k, t: O.common(k, 4, t, tax), "set"),
        "merge": (lambda k, t: O.merge_k(k, t, mode=O.PLAIN, tax=tax), "set"),
    }
    for name, (fn, kind) in ops.items():
        wk, wt = fn(files, taxs)
        gk, gt = by_ranges(fn, files, taxs, nranges=37, kind=kind, threads=4)
        assert np.array_equal(gk, wk) and np.array_equal(gt, wt), name
    # plain keys, and an inter whose third file is EMPTY: the loop stops there and keeps the running result
    with_empty = [files[0], files[2], np.empty(0, np.uint64), files[4]]
    want = O.inter(with_empty)
    assert len(want) > 0
    assert np.array_equal(by_ranges(lambda k, t: O.inter(k), with_empty, None, nranges=16, kind="inter", threads=3), want)
    assert np.array_equal(by_ranges(lambda k, t: O.union(k), files, None, nranges=16, threads=3), O.union(files))
