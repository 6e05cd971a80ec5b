"""The single-pass many-stream merge / union (csrc/ukm_srmerge.hip: one LDS tile per value range, stable in-LDS merge
sort) against the CPU oracle: mergeChunksFile's heap merge in all of its modes (util-sort.go:227-606: equal codes leave
in stream order; -u with the LCA fold; -d; the chunk protocol of a non-final round), the n-file `union` with its TaxId
fold (union.go:186-208) and `common` below the full threshold (common.go:220-344), on hundreds to 1024 streams.

UKM_SRMERGE=1 forces the route at test sizes (the library takes it for records with taxids from 512 streams and 2^24
records on);
`ctx.last_route() == 4` shows it answered.  UKM_SRMERGE_FILL over-fills the value ranges so that every range needs
several passes by value (the quota rule)."""
import numpy as np
import pytest

from conftest import splitmix64, synth_tree

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572
ROUTE_SR = 4
ROUTE_SR_COMMON = 5


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


_shapes = {}


def _shape(nfiles, per, p, T):
    """the files of one (nfiles, per, p) shape, drawn once for all its knob settings (700 draws over a universe of 3e6
    codes are 10 s of numpy per call: 40 s of the round-5 suite)"""
    key = (nfiles, per, p, T)
    if key not in _shapes:
        U = _universe(int(per / p))
        files = [U[_member(len(U), f, p, 7)] for f in range(nfiles)]
        files = [f for f in files if len(f)]
        _shapes[key] = (files, [_taxids(f + np.uint64(i), T, i) for i, f in enumerate(files)])
    return _shapes[key]


@pytest.mark.parametrize("buckets", [None, "1", "0", "2"])
@pytest.mark.parametrize("nfiles,per,p", [(200, 3000, 0.02), (1000, 700, 0.002), (1024, 300, 0.5), (65, 20000, 0.3), (700, 2500, 0.0008)])
def test_merge_modes_and_union_many_streams(env, monkeypatch, nfiles, per, p, buckets):
    """every mode of the merge and the union, plain and with taxids; files that hardly overlap, that overlap heavily
    (runs of hundreds of equal codes across files), 1024 = the most streams the route takes.  buckets: the tiles ordered by
    the library's own choice, by counting placement wherever a tile allows it (tiles with a crowded bucket fall back to the
    merge rounds one by one: both orders in one launch), by the merge rounds alone, by the dense code numbers wherever a
    tile holds at most 256 distinct codes (the other tiles are put back for the merge rounds)"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    if buckets is not None:
        monkeypatch.setenv("UKM_SRMERGE_BUCKETS", buckets)
    files, taxs = _shape(nfiles, per, p, T)
    gk, gt = ctx.merge_k(files, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_SR
    ek, et = _stable(files, taxs)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    assert np.array_equal(ctx.merge_k(files, mode=L.PLAIN), ek)
    assert ctx.last_route() == ROUTE_SR
    for mode in (L.UNIQUE, L.REPEATED):
        for final in (True, False):
            assert np.array_equal(ctx.merge_k(files, mode=mode, final_round=final),
                                  O.merge_k(files, mode=mode, final_round=final)), (mode, final)
            assert ctx.last_route() == ROUTE_SR
            gk, gt = ctx.merge_k(files, taxs, mode=mode, final_round=final)
            ok, ot = O.merge_k(files, taxs, mode=mode, final_round=final, tax=tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (mode, final)
    assert np.array_equal(ctx.union(files), O.union(files))
    assert ctx.last_route() == ROUTE_SR
    gk, gt = ctx.union(files, taxs)
    assert ctx.last_route() == ROUTE_SR
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    thr = max(2, int(nfiles * p * 0.8))
    gk, gt = ctx.common(files, thr, taxs)
    ok, ot = O.common(files, thr, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # the multi-level merge gives the same (the route is a choice, not a semantic)
    monkeypatch.setenv("UKM_SRMERGE", "0")
    gk2, gt2 = ctx.merge_k(files, taxs, mode=L.PLAIN)
    assert ctx.last_route() != ROUTE_SR
    assert np.array_equal(gk2, ek) and np.array_equal(gt2, et)


@pytest.mark.parametrize("clade", ["1", "0", None])
@pytest.mark.parametrize("nfiles,per,p", [(300, 2500, 0.02), (900, 900, 0.002), (600, 400, 0.4)])
def test_union_emit_with_clade_codes(env, monkeypatch, nfiles, per, p, clade):
    """Round 5: the single pass's `union` / `common` emit folding one-byte CLADE codes (UKM_SRMERGE_CLADE=1), pre-order numbers
    (0), or whichever its taxid sample picks (unset), against the oracle: taxids over the whole tree (unrelated: runs span
    clades), from one small clade (related: every run goes through the second, exact pass), with zeros and ids beyond the
    taxonomy (a run with one of them and another taxid folds to 0), and all records of a code carrying the same taxid."""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    if clade is None:
        monkeypatch.delenv("UKM_SRMERGE_CLADE", raising=False)
    else:
        monkeypatch.setenv("UKM_SRMERGE_CLADE", clade)
    U = _universe(int(per / p))
    files = [U[_member(len(U), f, p, 17)] for f in range(nfiles)]
    files = [f for f in files if len(f)]
    shapes = {
        "unrelated": [_taxids(f + np.uint64(i), T, i) for i, f in enumerate(files)],
        "related": [(np.uint64(T - 40) + splitmix64(np.uint64(SEED + i) ^ f) % np.uint64(40)).astype(np.uint32) for i, f in enumerate(files)],
        "by_code": [_taxids(f, T, 0) for f in files],            # (every record of a code carries the same taxid)
    }
    holes = [t.copy() for t in shapes["unrelated"]]
    for i, t in enumerate(holes):
        t[i % 5::11] = 0
        t[(i + 3) % 7::13] = np.uint32(T + 100 + i % 3)
    shapes["zeros_and_unknown"] = holes
    for name, taxs in shapes.items():
        gk, gt = ctx.union(files, taxs)
        assert ctx.last_route() == ROUTE_SR
        ok, ot = O.union(files, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (name, clade)
        thr = max(2, int(len(files) * p * 0.8))
        gk, gt = ctx.common(files, thr, taxs)
        ok, ot = O.common(files, thr, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (name, clade, "common")


@pytest.mark.parametrize("order", [None, "2"])
@pytest.mark.parametrize("fill", ["150", "400"])
def test_ranges_that_do_not_fit_a_tile_are_worked_off_by_value(env, monkeypatch, fill, order):
    """UKM_SRMERGE_FILL = 150 / 400 % of a tile per range on average: every range takes two to five passes (quota rule);
    multiset streams and ties across many streams included"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    monkeypatch.setenv("UKM_SRMERGE_FILL", fill)
    if order is not None:
        monkeypatch.setenv("UKM_SRMERGE_BUCKETS", order)
    rng = np.random.default_rng(int(fill))
    nfiles = 300
    streams = [np.sort(rng.integers(0, 1 << 20, 900 + 7 * i).astype(np.uint64)) for i in range(nfiles)]   # many ties, duplicates inside
    taxs = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams)]
    gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_SR
    ek, et = _stable(streams, taxs)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    gk, gt = ctx.union(streams, taxs)
    assert ctx.last_route() == ROUTE_SR
    ok, ot = O.union(streams, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.merge_k(streams, taxs, mode=L.REPEATED)
    ok, ot = O.merge_k(streams, taxs, mode=O.REPEATED, tax=tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


@pytest.mark.parametrize("order", [None, "2"])
def test_uneven_streams_all_ones_codes_and_a_crowded_code(env, monkeypatch, order):
    """stream sizes from 1 record to 60 000; real 2^64-1 codes (the kernel's own sentinel value); one code present in
    every stream several times (a run longer than a thread's share, shorter than a tile); then the same with one code in
    MORE copies than a tile holds: the route declines and the multi-level merge answers, same result"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    if order is not None:
        monkeypatch.setenv("UKM_SRMERGE_BUCKETS", order)
    rng = np.random.default_rng(9)
    sizes = [1, 2, 3, 60_000, 17, 9, 4608, 4609, 512, 1, 30_000] + [int(x) for x in rng.integers(1, 400, 190)]
    allones = np.uint64(0xFFFFFFFFFFFFFFFF)
    crowd = np.uint64(123_456_789)
    streams = []
    for i, n in enumerate(sizes):
        s = rng.integers(0, 1 << 62, n).astype(np.uint64)
        if i % 3 == 0:
            s = np.concatenate([s, np.full(1 + i % 4, allones)])
        s = np.concatenate([s, np.full(1 + i % 5, crowd)])
        streams.append(np.sort(s))
    taxs = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams)]
    gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_SR
    ek, et = _stable(streams, taxs)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    assert np.array_equal(ctx.merge_k(streams, mode=L.PLAIN), ek)
    gk, gt = ctx.union(streams, taxs)
    assert ctx.last_route() == ROUTE_SR
    ok, ot = O.union(streams, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    assert np.array_equal(ctx.union(streams), O.union(streams))
    # one code 40 times in each of 200 streams: 8000 copies > one tile
    streams2 = [np.sort(np.concatenate([s, np.full(40, crowd)])) for s in streams[:200]]
    taxs2 = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams2)]
    gk, gt = ctx.merge_k(streams2, taxs2, mode=L.PLAIN)
    assert ctx.last_route() != ROUTE_SR
    ek, et = _stable(streams2, taxs2)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    gk, gt = ctx.union(streams2, taxs2)
    ok, ot = O.union(streams2, taxs2, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_unsorted_stream_sends_the_call_to_the_general_route(env, monkeypatch):
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    rng = np.random.default_rng(2)
    streams = [np.sort(rng.integers(0, 1 << 40, 2000).astype(np.uint64)) for _ in range(120)]
    streams[77] = streams[77][::-1].copy()
    assert np.array_equal(ctx.merge_k(streams, mode=L.PLAIN), np.sort(np.concatenate(streams)))
    assert ctx.last_route() != ROUTE_SR
    assert np.array_equal(ctx.union(streams), O.union(streams))


def test_union_taxid_fold_on_a_forest_with_merged_zero_and_unknown_ids(monkeypatch):
    """the run fold of the union goes through pre-order numbers (smallest / largest per run, one table LCA); the reference
    folds LCA(taxid, lca) record by record (union.go:195-201).  Same answers on a forest of three trees, merged ids, taxid
    0 and unknown ids, runs where every member carries the same awkward id"""
    from oracle import oracle as O
    from unikmer_amd import lib as L
    monkeypatch.setenv("UKM_SRMERGE", "1")
    c = L.Context(0)
    child, parent = [], []
    child.append(100); parent.append(100)
    for i in range(101, 160):
        child.append(i); parent.append(i - 1)
    for i in range(100, 160, 5):
        child.append(1000 + i); parent.append(i)
    for t in range(1, 122):
        child.append(4999 + t); parent.append(4999 + (1 if t == 1 else (t - 2) // 3 + 1))
    child += [9001, 9002]; parent += [9000, 9001]
    child, parent = np.array(child, np.uint32), np.array(parent, np.uint32)
    mo, mn = np.array([50, 51, 52], np.uint32), np.array([159, 5003, 77777], np.uint32)
    c.taxonomy_load(child, parent, mo, mn)
    tax = O.Taxonomy(child, parent, mo, mn)
    rng = np.random.default_rng(23)
    pool = np.concatenate([child, [0, 50, 51, 52, 9000, 400, 99999]]).astype(np.uint32)
    U = _universe(5_000)
    nfiles = 80
    files = [U[_member(len(U), f, 0.25, 31)] for f in range(nfiles)]
    theme = rng.integers(0, 4, len(U))
    fixed = rng.choice(pool, len(U))
    pos = {int(code): i for i, code in enumerate(U)}
    taxs = []
    for f in range(nfiles):
        idx = np.array([pos[int(x)] for x in files[f]])
        t = rng.choice(pool, len(idx))
        same = theme[idx] == 0
        t[same] = fixed[idx][same]
        chain = theme[idx] == 1
        t[chain] = rng.integers(100, 160, int(chain.sum()))
        tern = theme[idx] == 2
        t[tern] = rng.integers(5000, 5121, int(tern.sum()))
        taxs.append(t.astype(np.uint32))
    gk, gt = c.union(files, taxs)
    assert c.last_route() == ROUTE_SR
    ok, ot = O.union(files, taxs, tax)
    assert len(ok) > 1000 and np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = c.merge_k(files, taxs, mode=L.REPEATED)
    ok, ot = O.merge_k(files, taxs, mode=O.REPEATED, tax=tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    c.close()


def test_default_choice_takes_the_route_for_many_streams(env):
    """without the knob: 600 streams x 30 000 records WITH taxids (1.8e7 >= 2^24) go through the single pass; the same
    streams without taxids, and 40 streams, take the multi-level merge (the library's choice follows the measurements)"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(5)
    streams = [np.sort(rng.integers(0, 1 << 61, 30_000).astype(np.uint64)) for _ in range(600)]
    taxs = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams)]
    gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_SR
    ek, et = _stable(streams, taxs)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    g = ctx.merge_k(streams, mode=L.PLAIN)
    assert ctx.last_route() != ROUTE_SR
    assert np.array_equal(g, ek)
    g = ctx.union(streams[:40])
    assert ctx.last_route() != ROUTE_SR
    assert np.array_equal(g, O.union(streams[:40]))


def test_streams_without_taxids_among_streams_with_taxids(env, monkeypatch):
    """files without TaxId information among files that carry it (a mixed `union` / `merge`: their records take part with
    taxid 0, union.go:144,195-201), 600 streams through the single pass"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    rng = np.random.default_rng(31)
    U = _universe(60_000)
    files = [U[_member(len(U), f, 0.05, 5)] for f in range(600)]
    files = [f for f in files if len(f)]
    taxs = [None if i % 7 == 3 else _taxids(f + np.uint64(i), T, i) for i, f in enumerate(files)]
    gk, gt = ctx.union(files, taxs)
    assert ctx.last_route() == ROUTE_SR
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.merge_k(files, taxs, mode=L.PLAIN)
    assert ctx.last_route() == ROUTE_SR
    zt = [np.zeros(len(f), np.uint32) if t is None else t for f, t in zip(files, taxs)]
    ek, et = _stable(files, zt)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)


@pytest.mark.parametrize("fill", [None, "400"])
def test_common_below_the_number_of_files_counts_inside_the_tiles(env, monkeypatch, fill):
    """common.go:220-344 with a threshold below the number of files: the single pass counts the records of every code and
    writes the codes that reach it (route 5), plain and with the TaxId fold; thresholds from 2 to the number of files;
    files with duplicates inside (the first file's count once: common.go:232,244, the others' every time); ranges of several
    passes (UKM_SRMERGE_FILL)"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_SRMERGE", "1")
    monkeypatch.setenv("UKM_COMMON_PROBE", "0")
    if fill:
        monkeypatch.setenv("UKM_SRMERGE_FILL", fill)
    nfiles, p = 300, 0.3
    U = _universe(20000)
    core = U[::7]
    files = []
    rng = np.random.default_rng(5)
    for f in range(nfiles):
        x = U[_member(len(U), f, p, 11)]
        if f % 3:
            x = np.union1d(x, core)                       # a core that most files hold
        if f % 50 == 0:
            x = np.sort(np.concatenate([x, x[::9]]))      # duplicates inside a file (the first file among them)
        files.append(x)
    taxs = [_taxids(x + np.uint64(i), T, i) for i, x in enumerate(files)]
    for thr in (2, 3, 60, 150, 199, 200, 201, nfiles - 1, nfiles):
        g = ctx.common(files, thr)
        assert ctx.last_route() == ROUTE_SR_COMMON, thr
        assert np.array_equal(g, O.common(files, thr)), thr
        gk, gt = ctx.common(files, thr, taxs)
        assert ctx.last_route() == ROUTE_SR_COMMON, thr
        ok, ot = O.common(files, thr, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), thr
    # some files without taxids (read as 0), and the multi-level route gives the same
    taxs2 = [t if i % 4 else None for i, t in enumerate(taxs)]
    gk, gt = ctx.common(files, 120, taxs2)
    assert ctx.last_route() == ROUTE_SR_COMMON
    ok, ot = O.common(files, 120, taxs2, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    monkeypatch.setenv("UKM_SRMERGE", "0")
    gk2, gt2 = ctx.common(files, 120, taxs2)
    assert ctx.last_route() != ROUTE_SR_COMMON
    assert np.array_equal(gk2, ok) and np.array_equal(gt2, ot)
