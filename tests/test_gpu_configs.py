"""BASELINE.json configs 3 and 4 at their real FILE COUNTS (100-stream merge tree with odd carries;
1000-link chained inter / diff fold with taxids, host peeks, early exits), bit-exact against the CPU
oracle at per-file sizes the oracle finishes in seconds.  Reference semantics: union.go:186-305,
util-sort.go:227-606 (mergeChunksFile), inter.go:205-286, diff.go:379-454.

Every case also runs with UKM_FORCE_TICKET=1 (the dispatch-order independent kernels)."""
import numpy as np
import pytest

from conftest import splitmix64, synth_tree

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572


@pytest.fixture(scope="module", params=["blockidx", "ticket"])
def env(request):
    import os
    from oracle import oracle as O
    from unikmer_amd import lib as L
    old = os.environ.get("UKM_FORCE_TICKET")
    if request.param == "ticket":
        os.environ["UKM_FORCE_TICKET"] = "1"   # read by ukm_ctx_create
    else:
        os.environ.pop("UKM_FORCE_TICKET", None)
    ctx = L.Context(0)
    if old is None:
        os.environ.pop("UKM_FORCE_TICKET", None)
    else:
        os.environ["UKM_FORCE_TICKET"] = old
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
    """membership draw of file f over a universe of n codes (independent draws per file)"""
    h = splitmix64(np.uint64(seed + 1000 * (f + 1)) ^ np.arange(n, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p


def _taxids(codes, T, salt):
    return (np.uint64(1) + splitmix64(np.uint64(SEED + 2 + salt) ^ codes) % np.uint64(T)).astype(np.uint32)


# ------------------------------------------------------------------------------------------- config 3
@pytest.mark.parametrize("nfiles", [100, 101, 37])
def test_config3_union_and_merge_many_files(env, nfiles):
    """union of 100 sorted files (p = 0.5 draws over one universe) and mergeChunksFile -u / -d / plain over
    the same streams; 101 and 37 files give odd carries at several tree levels."""
    O, L, ctx, tax, T = env
    U = _universe(40_000)
    files = [U[_member(len(U), f, 0.5, 11)] for f in range(nfiles)]
    assert np.array_equal(ctx.union(files), O.union(files))
    for mode in (L.UNIQUE, L.REPEATED, L.PLAIN):
        for final in (True, False):
            assert np.array_equal(ctx.merge_k(files, mode=mode, final_round=final),
                                  O.merge_k(files, mode=mode, final_round=final)), (mode, final)
    # with taxids: the LCA fold over up to `nfiles` occurrences of a code
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    gk, gt = ctx.union(files, taxs)
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.merge_k(files, taxs, mode=L.UNIQUE)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.merge_k(files, taxs, mode=L.REPEATED)
    ok, ot = O.merge_k(files, taxs, mode=O.REPEATED, tax=tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_config3_union_100_files_dirty_inputs(env):
    """the reference's hash-map union accepts anything (union.go:186-208): one multiset file, one unsorted
    file and one empty file among 100, at positions that land in different tree levels"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(3)
    U = _universe(30_000)
    files = [U[_member(len(U), f, 0.5, 29)] for f in range(100)]
    files[17] = np.sort(np.concatenate([files[17], files[17][:500]]))     # duplicates inside a file
    files[64] = rng.permutation(files[64])                                # unsorted
    files[99] = np.empty(0, np.uint64)                                    # empty (and the odd one out)
    files[3] = rng.permutation(np.concatenate([files[3], files[3][-50:]]))  # unsorted AND duplicated
    assert np.array_equal(ctx.union(files), O.union(files))
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    gk, gt = ctx.union(files, taxs)
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_config3_merge_uneven_stream_sizes(env):
    """sorted chunk files of very different sizes (a real `sort -m` leaves a short last chunk), some empty"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(5)
    sizes = [50_000, 0, 3, 20_000, 1, 9728, 9729, 0, 70_000, 64, 65, 19456, 5, 30_000, 12, 7, 40_000]
    streams = [np.sort(rng.integers(0, 1 << 30, n).astype(np.uint64)) for n in sizes]
    taxs = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams)]
    for mode in (L.PLAIN, L.UNIQUE, L.REPEATED):
        for final in (True, False):
            assert np.array_equal(ctx.merge_k(streams, mode=mode, final_round=final),
                                  O.merge_k(streams, mode=mode, final_round=final)), (mode, final)
            gk, gt = ctx.merge_k(streams, taxs, mode=mode, final_round=final)
            ok, ot = O.merge_k(streams, taxs, mode=mode, final_round=final, tax=tax)
            assert np.array_equal(gk, ok), (mode, final)
            if mode != L.PLAIN:   # plain: equal codes keep stream order in both, but check it explicitly below
                assert np.array_equal(gt, ot), (mode, final)
    # plain merge is stable: equal codes keep (stream, position) order
    gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
    cat = np.concatenate(streams)
    o = np.argsort(cat, kind="stable")
    assert np.array_equal(gk, cat[o]) and np.array_equal(gt, np.concatenate(taxs)[o])


# ------------------------------------------------------------------------------------------- config 4
def _chain_files(nfiles, n_universe, p, core_frac, seed, T):
    """config 4 shape: `nfiles` draws with probability p over one universe; a `core_frac` share of the
    universe is in EVERY file so that the 1000-fold intersection stays non-empty (p alone: 0.9^1000 = 0)."""
    U = _universe(n_universe, 24, SEED + seed)
    core = _member(n_universe, 10_000, core_frac, seed) if core_frac > 0 else np.zeros(n_universe, bool)
    files, taxs = [], []
    for f in range(nfiles):
        m = _member(n_universe, f, p, seed) | core
        files.append(U[m])
        taxs.append(_taxids(files[-1], T, f))
    return files, taxs


def _check_all(O, L, ctx, tax, files, taxs, expect_nonempty_inter=None):
    gk, gt = ctx.inter(files, taxs)
    ok, ot = O.inter(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    if expect_nonempty_inter is not None:
        assert (len(ok) > 0) == expect_nonempty_inter
    assert np.array_equal(ctx.inter(files), O.inter(files))
    gk, gt = ctx.inter(files, taxs, mix_taxid=True)
    ok, ot = O.inter(files, taxs, tax, mix_taxid=True)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.diff(files, taxs)
    ok, ot = O.diff(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    assert np.array_equal(ctx.diff(files), O.diff(files))
    gk, gt = ctx.diff(files, taxs, compare_taxid=True)
    ok, ot = O.diff(files, taxs, tax, compare_taxid=True)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_config4_chain_1000_files_nonempty(env):
    """inter / diff / diff -t over 1000 files with taxids: the running result survives all 999 links (every
    host peek at 4, 8, ..., 64, 128, ... sees a non-zero count; the small-link path clears its own status)"""
    O, L, ctx, tax, T = env
    files, taxs = _chain_files(1000, 2_400, 0.9, 0.15, 41, T)
    # give the first file private codes so that diff over 999 files is non-empty as well
    U2 = _universe(300, 24, SEED + 99) + np.uint64(1 << 40)
    files[0] = np.sort(np.concatenate([files[0], U2]))
    taxs[0] = _taxids(files[0], T, 0)
    _check_all(O, L, ctx, tax, files, taxs, expect_nonempty_inter=True)
    assert len(ctx.diff(files)) >= 300


def test_config4_chain_empties_midway(env):
    """p = 0.8 without a core: the intersection runs empty near link 40 (between two host peeks); the fold
    must stop there (inter.go:283-286) and diff must still visit every file"""
    O, L, ctx, tax, T = env
    files, taxs = _chain_files(1000, 2_500, 0.8, 0.0, 43, T)
    _check_all(O, L, ctx, tax, files, taxs, expect_nonempty_inter=False)


def test_config4_chain_empty_file_mid_chain(env):
    """an EMPTY later file: inter keeps the running result and stops (inter.go:211-217, flagBreak); diff
    skips it and goes on"""
    O, L, ctx, tax, T = env
    files, taxs = _chain_files(300, 2_400, 0.9, 0.2, 47, T)
    for pos in (150, 5, 299):
        f2, t2 = list(files), list(taxs)
        f2[pos] = np.empty(0, np.uint64)
        t2[pos] = np.empty(0, np.uint32)
        _check_all(O, L, ctx, tax, f2, t2)


def test_config4_chain_multi_tile_links(env):
    """links of several hundred tiles (first file 1.2e6 codes: the separate-memset, two-level partition path
    of ukm_dev_setop2_link) chained over 48 files with taxids"""
    O, L, ctx, tax, T = env
    files, taxs = _chain_files(48, 1_400_000, 0.9, 0.3, 53, T)
    _check_all(O, L, ctx, tax, files, taxs, expect_nonempty_inter=True)


def test_fold_shapes_tiny_first_file_and_dense_later_files(env):
    """shapes at the edge of the range fold's eligibility: a first file of a few records against much larger files (the
    fold would stream whole files through one workgroup: the chained per-file kernels answer), and later files several
    times denser than the first (slices of several chunks inside the fold)"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(77)
    U = np.cumsum(rng.integers(1, 1 << 20, 600_000).astype(np.uint64))
    big = [U[rng.random(len(U)) < 0.8] for _ in range(6)]
    tiny = np.sort(rng.choice(U, 40, replace=False))
    sparse = U[rng.random(len(U)) < 0.1]
    for first in (tiny, sparse):
        files = [first] + big
        taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
        for fn, ofn in ((ctx.inter, O.inter), (ctx.diff, O.diff)):
            gk, gt = fn(files, taxs)
            ok, ot = ofn(files, taxs, tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
            assert np.array_equal(fn(files), ofn(files))


def test_config4_chain_with_duplicates_falls_back(env):
    """a multiset file inside a long chain: the chained fold reports it and the synchronous fold (rank path,
    'equality advances both cursors') takes over"""
    O, L, ctx, tax, T = env
    files, taxs = _chain_files(40, 3_000, 0.9, 0.3, 59, T)
    files[20] = np.sort(np.concatenate([files[20], files[20][::7]]))
    taxs[20] = _taxids(files[20], T, 20)
    files[0] = np.sort(np.concatenate([files[0], files[0][::11]]))
    taxs[0] = _taxids(files[0], T, 0)
    for fn, ofn in ((ctx.inter, O.inter), (ctx.diff, O.diff)):
        gk, gt = fn(files, taxs)
        ok, ot = ofn(files, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
        assert np.array_equal(fn(files), ofn(files))


# ------------------------------------------------------------------------------------------- k-way kernel edges
@pytest.mark.parametrize("nstreams", [3, 4, 5, 8, 9, 17, 64, 65])
def test_kway_stream_counts_and_extreme_codes(env, nstreams, monkeypatch):
    """fan-in 4 (<= 4 streams) and 8, one / two / three levels, nodes with a single child, and the codes 0 and
    2^64-1 (the in-LDS merges use 2^64-1 as their sentinel: a real one takes the bounds-checked loop)"""
    O, L, ctx, tax, T = env
    monkeypatch.setenv("UKM_KWAY", "1")   # 3-4 tiny streams would take the pairwise tree by default (round 3)
    rng = np.random.default_rng(nstreams)
    streams = []
    for i in range(nstreams):
        n = int(rng.choice([0, 1, 7, 600, 5000, 40_000]))
        s = np.unique(rng.integers(0, 1 << 64, n, dtype=np.uint64))
        if i % 3 == 0:
            s = np.unique(np.concatenate([s, np.array([0, 2**64 - 1], dtype=np.uint64)]))
        if i % 5 == 1:
            s = np.unique(np.concatenate([s, np.array([2**64 - 1], dtype=np.uint64)]))
        streams.append(s)
    taxs = [_taxids(s, T, i) for i, s in enumerate(streams)]
    assert np.array_equal(ctx.union(streams), O.union(streams))
    gk, gt = ctx.union(streams, taxs)
    ok, ot = O.union(streams, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    for mode in (L.PLAIN, L.REPEATED):
        assert np.array_equal(ctx.merge_k(streams, mode=mode), O.merge_k(streams, mode=mode))
    gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
    cat, tcat = np.concatenate(streams), np.concatenate(taxs)
    o = np.argsort(cat, kind="stable")
    assert np.array_equal(gk, cat[o]) and np.array_equal(gt, tcat[o])
    thr = max(1, nstreams // 3)
    assert np.array_equal(ctx.common(streams, thr), O.common(streams, thr))


def test_kway_long_runs_and_multisets(env):
    """runs of one code longer than a chunk inside a stream (the k-way kernel gives up and the pairwise / sort route
    answers), shorter runs (handled in the kernel: all copies of a consumed code are in LDS), all streams equal"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(11)
    base = np.sort(rng.integers(0, 1 << 40, 30_000).astype(np.uint64))
    long_run = np.sort(np.concatenate([base, np.full(5000, base[100], dtype=np.uint64)]))
    short_runs = np.sort(np.concatenate([base, np.repeat(base[::50], 40)]))
    streams = [base, long_run, short_runs, base.copy(), np.sort(rng.integers(0, 1 << 40, 10).astype(np.uint64))]
    taxs = [_taxids(np.arange(len(s), dtype=np.uint64) + np.uint64(i), T, i) for i, s in enumerate(streams)]
    assert np.array_equal(ctx.union(streams), O.union(streams))
    gk, gt = ctx.union(streams, taxs)
    ok, ot = O.union(streams, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    assert np.array_equal(ctx.merge_k(streams, mode=L.PLAIN), np.sort(np.concatenate(streams)))
    assert np.array_equal(ctx.merge_k(streams, mode=L.REPEATED), O.merge_k(streams, mode=L.REPEATED))
    ssub = [short_runs, base, short_runs.copy()]          # no long run: stays in the k-way kernel
    gk, gt = ctx.merge_k(ssub, [taxs[2], taxs[0], taxs[2]], mode=L.PLAIN)
    cat, tcat = np.concatenate(ssub), np.concatenate([taxs[2], taxs[0], taxs[2]])
    o = np.argsort(cat, kind="stable")
    assert np.array_equal(gk, cat[o]) and np.array_equal(gt, tcat[o])
    same = [base] * 9
    assert np.array_equal(ctx.union(same), np.unique(base))
    assert np.array_equal(ctx.common(same, 9), np.unique(base))


# ------------------------------------------------------------------- union by LDS hash probes (ukm_punion.hip)
def test_probe_union_pipelined_steps_and_order_check(env, monkeypatch):
    """The plain probe pass as one software pipeline per wave (pu2_probe_kernel, round 6): its steps are 16-byte pairs, a
    slice that does not begin its file starts one record early, a last step of one record is moved back by one, lanes
    beyond the end re-read the last pair, the order check takes a pair's predecessor from the neighbouring lane.  Shapes
    that put every one of those cases at the edges: slices of 0 / 1 / 2 / 3 / 127 / 128 / 129 / 257 records at the start,
    in the middle and at the end of their files, later files of 0, 1 and 2 records, and -- the order check -- ONE swapped
    neighbouring pair anywhere in a later file (inside a lane's pair, between lanes, between steps, across a slice
    boundary, at either end of the file) must send the call to the other routes (union.go:186-208 has no order
    requirement: the result is the oracle's either way; what is asserted is that the probe pass noticed)."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(606)
    monkeypatch.setenv("UKM_PUNION", "2")
    # base: eight files over a universe of ~3 ranges of 2048 entries; later files built from chosen slice lengths
    U = _universe(6000, gap_bits=24)
    base = [U[_member(len(U), f, 0.9, 3)] for f in range(8)]
    bu = np.unique(np.concatenate(base))
    cut1, cut2 = bu[2048], bu[4096]                      # the ranges' first entries (PU_RANGE = 2048)
    lo_pool, mid_pool, hi_pool = U[U < cut1], U[(U >= cut1) & (U < cut2)], U[U >= cut2]
    later = []
    for a_, b_, c_ in ((0, 1, 0), (1, 0, 1), (2, 3, 1), (127, 128, 129), (128, 127, 1), (129, 257, 2), (257, 1, 128), (0, 0, 1),
                       (1, 0, 0), (3, 129, 127), (500, 640, 385), (256, 256, 256)):
        parts = [np.sort(rng.choice(pool, n, replace=False)) for pool, n in ((lo_pool, a_), (mid_pool, b_), (hi_pool, c_))]
        later.append(np.concatenate(parts).astype(np.uint64))
    later += [np.empty(0, np.uint64), U[77:78].copy(), U[100:102].copy(), np.array([bu[-1] + np.uint64(5)], np.uint64)]
    # private codes too (misses that are claimed / listed)
    later.append(np.unique(rng.integers(1, int(U[-1]), 700).astype(np.uint64)))
    files = base + later
    assert np.array_equal(ctx.union(files), O.union(files))
    assert ctx.last_route() == 3
    # one swapped neighbouring pair: every position class of a 900-record later file
    victim = np.sort(rng.choice(U, 900, replace=False)).astype(np.uint64)
    b1 = int(np.searchsorted(victim, cut1))               # first record of the second range's slice
    b2 = int(np.searchsorted(victim, cut2))
    spots = {0, 1, 2, 126, 127, 128, 129, 254, 255, 256, b1 - 2, b1 - 1, b1, b1 + 1, b1 + 126, b1 + 127, b2 - 1, b2, len(victim) - 2,
             len(victim) - 3} | set(int(x) for x in rng.integers(0, len(victim) - 1, 12))
    for sp in sorted(x for x in spots if 0 <= x < len(victim) - 1):
        v = victim.copy()
        v[sp], v[sp + 1] = v[sp + 1], v[sp]
        trial = base + [later[3], v, later[10]]
        assert np.array_equal(ctx.union(trial), O.union(trial)), sp
        assert ctx.last_route() != 3, sp                   # the probe pass saw the inversion and backed out
    clean = base + [later[3], victim, later[10]]
    assert np.array_equal(ctx.union(clean), O.union(clean))
    assert ctx.last_route() == 3


def test_probe_union_matches_oracle(env, monkeypatch):
    """`union` of many plain sets that overlap heavily: the first eight files become the base set, every later record
    is one hash probe in the LDS table of its range, misses are sorted and merged in (ukm_punion.hip; the reference
    probes a hash map per k-mer, union.go:186-208).  UKM_PUNION=1 takes the path whatever the size, =2 also without
    the hit-rate guard.  Shapes: config 3's draws over one universe (few misses), a base set of less than one range
    and of many, more later files than one launch holds, later files that share nothing with the base set (every
    record a miss), 64-bit hashes including all-ones codes, duplicates inside files; an unsorted later file and a
    first-eight file that is unsorted make the path back out and the general route answer."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(41)
    monkeypatch.setenv("UKM_PUNION", "1")
    # (p = 0.15 / 0.1: 73 % / 57 % of the later records are in the base set: hundreds of new codes per range are claimed)
    for n_univ, nfiles, p in ((60_000, 40, 0.5), (3_000, 30, 0.5), (200_000, 26, 0.35), (20_000, 600, 0.3), (100_000, 40, 0.15),
                              (30_000, 300, 0.1)):
        U = _universe(n_univ)
        files = [U[_member(len(U), f, p, 77)] for f in range(nfiles)]
        assert np.array_equal(ctx.union(files), O.union(files)), (n_univ, nfiles, p)
        assert ctx.last_route() == 3, (n_univ, nfiles, p)
    # later files with private codes (misses), duplicates inside files, all-ones hashes
    U = _universe(50_000, gap_bits=40)
    files = [U[_member(len(U), f, 0.6, 5)] for f in range(30)]
    extra = np.sort(rng.integers(0, 1 << 63, 3000, dtype=np.uint64) * np.uint64(2) + np.uint64(1))
    files[12] = np.sort(np.concatenate([files[12], extra[:2000]]))
    files[20] = np.sort(np.concatenate([files[20], extra[1000:], np.full(3, np.uint64(2**64 - 1))]))
    files[25] = np.sort(np.concatenate([files[25], files[25][:700]]))          # a multiset
    files[2] = np.sort(np.concatenate([files[2], np.full(2, np.uint64(2**64 - 1))]))   # all ones inside the base set
    assert np.array_equal(ctx.union(files), O.union(files))
    # nothing in common with the base set: the guard backs out (mode 1) / every record is a miss (mode 2)
    disjoint = [np.sort(rng.choice(1 << 40, 4000, replace=False).astype(np.uint64) + np.uint64(f << 44)) for f in range(20)]
    assert np.array_equal(ctx.union(disjoint), O.union(disjoint))
    monkeypatch.setenv("UKM_PUNION", "2")
    assert np.array_equal(ctx.union(disjoint), O.union(disjoint))
    assert np.array_equal(ctx.union(files), O.union(files))
    # thousands of new codes per range (the base set has three ranges): the LDS list of a range overflows into the
    # global chunks, and past 2048 claimed codes the rest is listed without being inserted (duplicates in the list)
    Us = _universe(5_000, gap_bits=30)
    small = [Us[_member(len(Us), f, 0.7, 9)] for f in range(8)]
    fresh = np.unique(rng.integers(0, int(Us[-1]), 20_000).astype(np.uint64))
    crowded = small + [np.sort(np.concatenate([Us[_member(len(Us), 8 + f, 0.7, 9)], fresh[_member(len(fresh), f, 0.6, 13)]]))
                       for f in range(22)]
    assert np.array_equal(ctx.union(crowded), O.union(crowded))
    # a dense private cluster behind the base set's last code: one range would have to stream it all, the load guard
    # backs out (mode 1) / the one workgroup does it (mode 2)
    Ub = _universe(1_000_000, gap_bits=20)
    big = [Ub[_member(len(Ub), f, 0.5, 21)] for f in range(8)]
    tail = Ub[-1] + np.uint64(1) + np.arange(400_000, dtype=np.uint64) * np.uint64(3)
    big += [np.concatenate([Ub[_member(len(Ub), 8 + f, 0.5, 21)], tail[_member(len(tail), f, 0.8, 23)]]) for f in range(12)]
    expect = np.unique(np.concatenate(big))
    monkeypatch.setenv("UKM_PUNION", "1")
    assert np.array_equal(ctx.union(big), expect)
    monkeypatch.setenv("UKM_PUNION", "2")
    assert np.array_equal(ctx.union(big), expect)
    # unsorted inputs: in the later files and among the first eight
    dirty = list(files)
    dirty[15] = rng.permutation(dirty[15])
    assert np.array_equal(ctx.union(dirty), O.union(dirty))
    dirty = list(files)
    dirty[3] = rng.permutation(dirty[3])
    assert np.array_equal(ctx.union(dirty), O.union(dirty))
    # with taxids: the same pass with the TaxId fold in the table (next test); UKM_PUNION_TAX=0 keeps them off it
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    ok, ot = O.union(files, taxs, tax)
    gk, gt = ctx.union(files, taxs)
    assert ctx.last_route() == 3
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    monkeypatch.setenv("UKM_PUNION_TAX", "0")
    gk, gt = ctx.union(files, taxs)
    assert ctx.last_route() != 3
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_probe_union_with_taxids_matches_oracle(env, monkeypatch):
    """union.go:195-201 — the TaxId of a code is the LCA over every record that carries it — through the hash-probe pass:
    every table entry keeps the TaxId it came with and the smallest / largest pre-order number of the records that differ
    from it, one table LCA per entry at the end.  Shapes: uniformly random taxids; one taxid per file (k-mers of one
    genome); files without taxids among files with; new codes (claimed with their fold), more new codes than a table
    claims (listed record by record), all-ones codes, duplicates inside files, more later files than one launch holds"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(43)
    monkeypatch.setenv("UKM_PUNION", "1")
    for n_univ, nfiles, p in ((60_000, 40, 0.5), (3_000, 30, 0.5), (200_000, 26, 0.35), (20_000, 600, 0.3)):
        U = _universe(n_univ)
        files = [U[_member(len(U), f, p, 78)] for f in range(nfiles)]
        for kind in ("random", "file", "some"):
            if kind == "random":
                taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
            elif kind == "file":
                taxs = [np.full(len(f), 1 + (i * 7919) % T, np.uint32) for i, f in enumerate(files)]
            else:
                taxs = [_taxids(f, T, i) if i % 3 else None for i, f in enumerate(files)]
            gk, gt = ctx.union(files, taxs)
            assert ctx.last_route() == 3, (n_univ, nfiles, kind)
            ok, ot = O.union(files, taxs, tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (n_univ, nfiles, kind)
    # tiny files in front (a plasmid before the genomes): the base set is built from the LARGEST files
    U = _universe(40_000)
    files = [U[_member(len(U), f, 0.5, 6)] for f in range(30)]
    for i in (0, 1, 2, 5):
        files[i] = files[i][:7 + i]
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    assert np.array_equal(ctx.union(files), O.union(files)) and ctx.last_route() == 3
    gk, gt = ctx.union(files, taxs)
    assert ctx.last_route() == 3
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # later files with private codes, duplicates inside files (their taxids differ), all-ones hashes
    U = _universe(50_000, gap_bits=40)
    files = [U[_member(len(U), f, 0.6, 5)] for f in range(30)]
    extra = np.sort(rng.integers(0, 1 << 63, 3000, dtype=np.uint64) * np.uint64(2) + np.uint64(1))
    files[12] = np.sort(np.concatenate([files[12], extra[:2000]]))
    files[20] = np.sort(np.concatenate([files[20], extra[1000:], np.full(3, np.uint64(2**64 - 1))]))
    files[25] = np.sort(np.concatenate([files[25], files[25][:700]]))
    files[2] = np.sort(np.concatenate([files[2], np.full(2, np.uint64(2**64 - 1))]))
    taxs = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in files]
    ok, ot = O.union(files, taxs, tax)
    for mode in ("1", "2"):
        monkeypatch.setenv("UKM_PUNION", mode)
        gk, gt = ctx.union(files, taxs)
        assert ctx.last_route() == 3
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), mode
    # thousands of new codes per range: past the table's share the rest is listed record by record
    Us = _universe(5_000, gap_bits=30)
    small = [Us[_member(len(Us), f, 0.7, 9)] for f in range(8)]
    fresh = np.unique(rng.integers(0, int(Us[-1]), 20_000).astype(np.uint64))
    crowded = small + [np.sort(np.concatenate([Us[_member(len(Us), 8 + f, 0.7, 9)], fresh[_member(len(fresh), f, 0.6, 13)]]))
                       for f in range(22)]
    taxs = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in crowded]
    gk, gt = ctx.union(crowded, taxs)
    assert ctx.last_route() == 3
    ok, ot = O.union(crowded, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # disjoint files (every record a miss) and an unsorted later file
    disjoint = [np.sort(rng.choice(1 << 40, 4000, replace=False).astype(np.uint64) + np.uint64(f << 44)) for f in range(20)]
    taxs = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in disjoint]
    gk, gt = ctx.union(disjoint, taxs)
    ok, ot = O.union(disjoint, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    dirty = list(files)
    taxs = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in dirty]
    perm = rng.permutation(len(dirty[15]))
    dirty[15], taxs[15] = dirty[15][perm], taxs[15][perm]
    gk, gt = ctx.union(dirty, taxs)
    assert ctx.last_route() != 3
    ok, ot = O.union(dirty, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_common_by_counting_probes_matches_oracle(env, monkeypatch):
    """common.go:220-344 with a threshold below the number of files through the counting tables of the probe pass (route 6):
    the first file's codes count once (duplicates in it collapse, the last TaxId wins), every record of the other files
    counts (duplicates inside them twice), codes the first file lacks are claimed with their count and fold; plain, random
    taxids, one taxid per file, files without taxids; thresholds from 2 to the number of files; inputs the path declines
    (an unsorted later file, all-ones codes, files that share little with the first) give the same result another way"""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(47)
    monkeypatch.setenv("UKM_PUNION", "1")
    monkeypatch.setenv("UKM_COMMON_PROBE", "0")
    # (p = 0.35 / 0.3: the first file alone holds too little of the others, the base set is the union of the first four and
    #  every file is probed)
    for n_univ, nfiles, p in ((60_000, 40, 0.7), (3_000, 30, 0.8), (200_000, 26, 0.6), (20_000, 300, 0.75), (50_000, 40, 0.35), (8_000, 200, 0.3)):
        U = _universe(n_univ)
        files = [U[_member(len(U), f, p, 79)] for f in range(nfiles)]
        files[3] = np.sort(np.concatenate([files[3], files[3][::5]]))       # duplicates inside a later file count twice
        files[0] = np.sort(np.concatenate([files[0], files[0][::7]]))       # ... inside the first file once
        for kind in ("plain", "random", "file", "some"):
            if kind == "plain":
                taxs = None
            elif kind == "random":
                taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
            elif kind == "file":
                taxs = [np.full(len(f), 1 + (i * 7919) % T, np.uint32) for i, f in enumerate(files)]
            else:
                taxs = [_taxids(f, T, i) if i % 3 else None for i, f in enumerate(files)]
            # `merge -d` in its final round (util-sort.go:519-530): the codes with at least two records, every record of
            # every file counted (the duplicates inside the first file too) -- the same tables with a threshold of two
            if taxs is None:
                assert np.array_equal(ctx.merge_k(files, mode=L.REPEATED), O.merge_k(files, mode=O.REPEATED)), (n_univ, nfiles, kind)
                assert ctx.last_route() == 6, (n_univ, nfiles, kind)
            else:
                gk, gt = ctx.merge_k(files, taxs, mode=L.REPEATED)
                assert ctx.last_route() == 6, (n_univ, nfiles, kind)
                ok, ot = O.merge_k(files, taxs, mode=O.REPEATED, tax=tax)
                assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (n_univ, nfiles, kind)
            for thr in (2, nfiles // 2, nfiles - 1, nfiles, nfiles + 1):
                if taxs is None:
                    g = ctx.common(files, thr)
                    assert ctx.last_route() == 6, (n_univ, nfiles, kind, thr)
                    assert np.array_equal(g, O.common(files, thr)), (n_univ, nfiles, kind, thr)
                else:
                    gk, gt = ctx.common(files, thr, taxs)
                    assert ctx.last_route() == 6, (n_univ, nfiles, kind, thr)
                    ok, ot = O.common(files, thr, taxs, tax)
                    assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (n_univ, nfiles, kind, thr)
    # more new codes than a table claims, all-ones codes, an unsorted later file, files that share nothing: declined
    U = _universe(30_000)
    files = [U[_member(len(U), f, 0.7, 81)] for f in range(30)]
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    want = O.common(files, 20, taxs, tax)
    cases = {}
    extra = np.unique(rng.integers(0, int(U[-1]), 80_000).astype(np.uint64))
    cases["crowded"] = [f if i < 8 else np.union1d(f, extra[_member(len(extra), i, 0.8, 83)]) for i, f in enumerate(files)]
    cases["allones"] = [np.concatenate([f, np.full(2, np.uint64(2**64 - 1))]) if i in (0, 9, 12) else f for i, f in enumerate(files)]
    for name, fs in cases.items():
        ts = [_taxids(f, T, i) for i, f in enumerate(fs)]
        gk, gt = ctx.common(fs, 3, ts)
        ok, ot = O.common(fs, 3, ts, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), name
    dirty, dt = list(files), list(taxs)
    perm = rng.permutation(len(dirty[15]))
    dirty[15], dt[15] = dirty[15][perm], dt[15][perm]
    gk, gt = ctx.common(dirty, 20, dt)
    assert ctx.last_route() != 6
    assert np.array_equal(gk, want[0]) and np.array_equal(gt, want[1])
    disjoint = [np.sort(rng.choice(1 << 40, 4000, replace=False).astype(np.uint64) + np.uint64(f << 44)) for f in range(26)]
    assert len(ctx.common(disjoint, 2)) == 0 and ctx.last_route() != 6
    monkeypatch.setenv("UKM_PUNION", "0")
    gk, gt = ctx.common(files, 20, taxs)
    assert ctx.last_route() != 6
    assert np.array_equal(gk, want[0]) and np.array_equal(gt, want[1])


def test_probe_union_taxid_fold_on_a_forest_with_merged_zero_and_unknown_ids(monkeypatch):
    """the fold in the probe table against the reference's record-by-record LCA on a forest of three trees, merged ids,
    taxid 0, unknown ids (among them 2^32 - 2; 2^32 - 1 is the table's own marker: the call takes the general route),
    codes whose every record carries the same awkward id"""
    from oracle import oracle as O
    from unikmer_amd import lib as L
    monkeypatch.setenv("UKM_PUNION", "2")
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
    rng = np.random.default_rng(29)
    pool = np.concatenate([child, [0, 50, 51, 52, 9000, 400, 99999, 2**32 - 2]]).astype(np.uint32)
    U = _universe(5_000)
    nfiles = 60
    files = [U[_member(len(U), f, 0.4, 37)] for f in range(nfiles)]
    files[30] = np.sort(np.concatenate([files[30], U[-1] + np.uint64(5) + np.arange(300, dtype=np.uint64)]))   # new codes
    files[31] = np.sort(np.concatenate([files[31], U[-1] + np.uint64(5) + np.arange(0, 300, 2, dtype=np.uint64)]))
    theme = {}
    taxs = []
    for f in range(nfiles):
        t = rng.choice(pool, len(files[f]))
        for i, x in enumerate(files[f]):
            th = theme.setdefault(int(x), (int(rng.integers(0, 4)), int(rng.choice(pool))))
            if th[0] == 0:
                t[i] = th[1]                              # every record of the code carries the same id
            elif th[0] == 1:
                t[i] = rng.integers(100, 160)             # one chain
            elif th[0] == 2:
                t[i] = rng.integers(5000, 5121)           # one ternary tree
        taxs.append(t.astype(np.uint32))
    gk, gt = c.union(files, taxs)
    assert c.last_route() == 3
    ok, ot = O.union(files, taxs, tax)
    assert len(ok) > 1000 and np.array_equal(gk, ok) and np.array_equal(gt, ot)
    taxs[40][7] = np.uint32(2**32 - 1)
    gk, gt = c.union(files, taxs)
    assert c.last_route() != 3
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    c.close()


# ---------------------------------------------------------- inter / diff by LDS hash probes (ukm_pfold.hip)
def test_probe_fold_step_edges_exactly_once_and_strict_order(env, monkeypatch):
    """The streaming skeleton of round 6 inside the probe fold (pf_probe_kernel): `inter` COUNTS hits, so every record of a
    later file must be seen exactly once whatever its slice's length -- 0 / 1 / 2 / 3 / 127 / 128 / 129 / 255 / 257 records
    at the start, in the middle and at the end of a file, files of 1 and 2 records -- and every file must be STRICTLY
    increasing: one duplicated or one swapped neighbouring pair at every position class of a later file (inside a lane's
    pair, between lanes, between steps, across a range boundary, at either end) must reach the exact routes and still
    give the oracle's answer (inter.go:205-286, diff.go:379-454 incl. -t; a duplicate changes `inter`'s multiset result).
    First file of two ranges (L <= 1536 / 2048), taxids on every file."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(6061)
    monkeypatch.setenv("UKM_PFOLD_TAX", "1")
    U = _universe(4000, gap_bits=24)
    first = U[::2][:1800].copy()                          # 1800 records: two ranges of the inter-with-taxids variant (L = 900)
    cut = first[900]
    lo_pool, hi_pool = U[U < cut], U[U >= cut]
    later = []
    for a_, b_ in ((0, 1), (1, 0), (2, 3), (127, 128), (128, 129), (129, 127), (255, 257), (257, 1), (1, 1), (3, 2), (640, 513)):
        later.append(np.concatenate([np.sort(rng.choice(lo_pool, a_, replace=False)), np.sort(rng.choice(hi_pool, b_, replace=False))]).astype(np.uint64))
    later += [first[5:6].copy(), first[10:12].copy()]
    files = [first] + later
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]

    def check(fs, ts):
        for got, want in ((ctx.inter(fs, ts), O.inter(fs, ts, tax)), (ctx.diff(fs, ts), O.diff(fs, ts, tax)),
                          (ctx.diff(fs, ts, compare_taxid=True), O.diff(fs, ts, tax, compare_taxid=True))):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(ctx.inter(fs), O.inter(fs)) and np.array_equal(ctx.diff(fs), O.diff(fs))
    check(files, taxs)
    # files that all hold a common core (a non-empty intersection whose hit counts must come out exact) in slices of every length
    core = np.sort(rng.choice(first, 300, replace=False))
    for n in (1, 2, 127, 128, 129, 255, 256, 257, 511, 900):
        fs = [first] + [np.unique(np.concatenate([core, np.sort(rng.choice(U, n, replace=False))])).astype(np.uint64) for _ in range(5)]
        ts = [_taxids(f, T, 100 + i) for i, f in enumerate(fs)]
        got, want = ctx.inter(fs, ts), O.inter(fs, ts, tax)
        assert len(want[0]) >= 300 and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), n
    # one duplicated / one swapped neighbouring pair in a later file of 700 records
    victim = np.unique(np.concatenate([core, np.sort(rng.choice(U, 500, replace=False))])).astype(np.uint64)
    b1 = int(np.searchsorted(victim, cut))
    spots = sorted({0, 1, 2, 126, 127, 128, 129, 254, 255, 256, b1 - 2, b1 - 1, b1, b1 + 1, len(victim) - 2, len(victim) - 3} |
                   set(int(x) for x in rng.integers(0, len(victim) - 1, 8)))
    others = [np.unique(np.concatenate([core, np.sort(rng.choice(U, 400, replace=False))])).astype(np.uint64) for _ in range(4)]
    for sp in (x for x in spots if 0 <= x < len(victim) - 1):
        for kind in ("dup", "swap"):
            v = victim.copy()
            if kind == "dup":
                v[sp + 1] = v[sp]
            else:
                v[sp], v[sp + 1] = v[sp + 1], v[sp]
            fs = [first] + others[:2] + [v] + others[2:]
            ts = [_taxids(f, T, 200 + i) for i, f in enumerate(fs)]
            if kind == "swap":   # an unsorted stream is an error of these operations (as in the other routes)
                with pytest.raises(L.UnsortedError):
                    ctx.inter(fs, ts)
                continue
            got, want = ctx.inter(fs, ts), O.inter(fs, ts, tax)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (sp, kind)
            got, want = ctx.diff(fs, ts), O.diff(fs, ts, tax)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (sp, kind)


def test_probe_fold_shapes(env, monkeypatch):
    """`inter` (plain, LCA of taxids) and `diff` (plain) over many files through the hash-probe fold: same answers as the
    oracle's sequential folds (inter.go:205-286, diff.go:379-454) AND as the range fold of ukm_fold.hip (UKM_NO_PFOLD=1),
    for later files with and without taxids, one-record files, an all-ones code in the first file or in a later one
    (the table's empty marker: exact route), first files of one range and of many, an unsorted stream (error)."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(91)

    monkeypatch.setenv("UKM_PFOLD_TAX", "1")   # (the default: inter with taxids through the probe fold as well)

    def both(files, taxs):
        res = []
        for no_pf in (None, "1"):
            if no_pf is None:
                monkeypatch.delenv("UKM_NO_PFOLD", raising=False)
            else:
                monkeypatch.setenv("UKM_NO_PFOLD", no_pf)
            res.append((ctx.inter(files, taxs), ctx.inter(files), ctx.diff(files, taxs), ctx.diff(files),
                        ctx.diff(files, taxs, compare_taxid=True)))
        monkeypatch.delenv("UKM_NO_PFOLD", raising=False)
        (ik, it), i0, (dk, dt), d0, (ck, ct) = res[0]
        (ik2, it2), i02, (dk2, dt2), d02, (ck2, ct2) = res[1]
        assert np.array_equal(ik, ik2) and np.array_equal(it, it2) and np.array_equal(i0, i02)
        assert np.array_equal(dk, dk2) and np.array_equal(dt, dt2) and np.array_equal(d0, d02)
        assert np.array_equal(ck, ck2) and np.array_equal(ct, ct2)
        ok, ot = O.diff(files, taxs, tax, compare_taxid=True)
        assert np.array_equal(ck, ok) and np.array_equal(ct, ot)
        ok, ot = O.inter(files, taxs, tax)
        assert np.array_equal(ik, ok) and np.array_equal(it, ot) and np.array_equal(i0, O.inter(files))
        ok, ot = O.diff(files, taxs, tax)
        assert np.array_equal(dk, ok) and np.array_equal(dt, ot) and np.array_equal(d0, O.diff(files))

    for n_univ, nfiles, p, core in ((3_000, 40, 0.9, 0.2), (700_000, 12, 0.9, 0.3), (200, 9, 0.7, 0.3), (60_000, 130, 0.95, 0.1)):
        files, taxs = _chain_files(nfiles, n_univ, p, core, 61, T)
        both(files, taxs)
    files, taxs = _chain_files(30, 20_000, 0.9, 0.25, 67, T)
    # a later file of ONE record that is in the core / that is not; later files whose taxids are all zero
    core_code = O.inter(files)[:1]
    f2, t2 = list(files), list(taxs)
    f2[7], t2[7] = core_code.copy(), _taxids(core_code, T, 7)
    both(f2, t2)
    f2[9], t2[9] = np.array([3], np.uint64), np.array([5], np.uint32)
    both(f2, t2)
    f2, t2 = list(files), list(taxs)
    t2[4] = np.zeros(len(f2[4]), np.uint32)
    t2[11] = np.zeros(len(f2[11]), np.uint32)
    both(f2, t2)
    # diff -t keeps a code whose taxid in the later file equals the first file's or lies below it: later files that
    # copy the first file's taxids for the codes they share (kept), and files with the PARENT-side taxid (removed)
    f2, t2 = list(files), list(taxs)
    first = dict(zip(files[0].tolist(), taxs[0].tolist()))
    for j in (3, 8, 21):
        t2[j] = np.array([first.get(int(c), int(t)) for c, t in zip(f2[j], t2[j])], np.uint32)
    both(f2, t2)
    # all-ones codes: in the first file (and everywhere: it survives inter), in later files only
    ones = np.array([2**64 - 1], np.uint64)
    f2 = [np.concatenate([f, ones]) for f in files]
    t2 = [np.concatenate([t, np.array([7], np.uint32)]) for t in taxs]
    both(f2, t2)
    f2 = [files[0]] + [np.concatenate([f, ones]) for f in files[1:]]
    t2 = [taxs[0]] + [np.concatenate([t, np.array([7], np.uint32)]) for t in taxs[1:]]
    both(f2, t2)
    # a duplicate inside a later file / inside the first: the exact multiset route
    f2, t2 = list(files), list(taxs)
    f2[13] = np.sort(np.concatenate([f2[13], f2[13][:5]]))
    t2[13] = _taxids(f2[13], T, 13)
    both(f2, t2)
    f2, t2 = list(files), list(taxs)
    f2[0] = np.sort(np.concatenate([f2[0], f2[0][-3:]]))
    t2[0] = _taxids(f2[0], T, 0)
    both(f2, t2)
    # an unsorted later stream is an error for inter (the reference's 2-pointer walk would silently miss codes)
    f2 = list(files)
    f2[5] = rng.permutation(f2[5])
    with pytest.raises(L.UkmError):
        ctx.inter(f2)


def test_probe_fold_inter_taxids_forest_merged_unknown(monkeypatch):
    """inter with taxids through the probe fold folds pre-order numbers (minimum / maximum per record) and does one
    table LCA per survivor; the reference folds LCA(LCA(t0, t1), t2) ... file by file.  Same answers on the awkward
    ids: a forest of three trees (LCA across trees = 0), merged ids (resolved, except when every taxid of a record is the
    same old id), taxid 0 and unknown ids (0, unless every taxid of the record is that same id), deep chains."""
    from oracle import oracle as O
    from unikmer_amd import lib as L
    c = L.Context(0)
    child, parent = [], []
    child.append(100); parent.append(100)
    for i in range(101, 160):
        child.append(i); parent.append(i - 1)                  # a chain of 60
    for i in range(100, 160, 5):
        child.append(1000 + i); parent.append(i)               # leaves off the chain
    for t in range(1, 122):                                     # a complete ternary tree of depth 4 at 5000
        child.append(4999 + t); parent.append(4999 + (1 if t == 1 else (t - 2) // 3 + 1))
    child += [9001, 9002]; parent += [9000, 9001]              # a root that only appears as a parent
    child, parent = np.array(child, np.uint32), np.array(parent, np.uint32)
    mo, mn = np.array([50, 51, 52], np.uint32), np.array([159, 5003, 77777], np.uint32)
    c.taxonomy_load(child, parent, mo, mn)
    tax = O.Taxonomy(child, parent, mo, mn)
    rng = np.random.default_rng(17)
    pool = np.concatenate([child, [0, 50, 51, 52, 9000, 400, 99999]]).astype(np.uint32)
    U = _universe(6_000)
    nfiles = 12
    files = [U[_member(len(U), f, 0.93, 31)] for f in range(nfiles)]
    taxs = []
    # per code a "theme": all files the same id (incl. 0 / merged / unknown), ids of one clade, or anything
    theme = rng.integers(0, 4, len(U))
    fixed = rng.choice(pool, len(U))
    pos = {int(code): i for i, code in enumerate(U)}
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
    for env_tax in ("1", "0"):   # the probe fold / the range fold
        monkeypatch.setenv("UKM_PFOLD_TAX", env_tax)
        gk, gt = c.inter(files, taxs)
        ok, ot = O.inter(files, taxs, tax)
        assert len(ok) > 1000
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), env_tax
    c.close()


def test_common_of_all_files_through_the_probe_fold(env, monkeypatch):
    """`common` with the default threshold (every file, common.go:93-105 with -p 1) over duplicate-free sorted files is
    answered by the hash-probe fold; same answers as the oracle's counting map (common.go:220-344) and as the counting
    merge (UKM_COMMON_PROBE=0).  A duplicate in a later file (counts twice there: a code missing elsewhere can still reach
    the threshold), duplicates in the first file (collapse), an empty file, an unsorted file, streams without taxids and
    the all-ones code must come out the same way through the fallback."""
    O, L, ctx, tax, T = env

    def both(files, taxs, thr=None):
        thr = len(files) if thr is None else thr
        res = []
        for knob in (None, "0"):
            if knob is None:
                monkeypatch.delenv("UKM_COMMON_PROBE", raising=False)
            else:
                monkeypatch.setenv("UKM_COMMON_PROBE", knob)
            res.append((ctx.common(files, thr, taxs), ctx.common(files, thr)))
        monkeypatch.delenv("UKM_COMMON_PROBE", raising=False)
        (gk, gt), g0 = res[0]
        (gk2, gt2), g02 = res[1]
        ok, ot = O.common(files, thr, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (len(gk), len(ok))
        assert np.array_equal(gk2, ok) and np.array_equal(gt2, ot)
        assert np.array_equal(g0, O.common(files, thr)) and np.array_equal(g02, g0)
        return ok

    for n_univ, nfiles, p, core in ((3_000, 40, 0.9, 0.2), (700_000, 12, 0.9, 0.3), (200, 9, 0.7, 0.3), (60_000, 130, 0.95, 0.1),
                                    (5_000, 4, 0.5, 0.0)):
        files, taxs = _chain_files(nfiles, n_univ, p, core, 71, T)
        ok = both(files, taxs)
        if core > 0:
            assert len(ok) > 0
        both(files, taxs, len(files) - 1)      # (another threshold: the counting merge)
    files, taxs = _chain_files(24, 20_000, 0.9, 0.25, 73, T)
    core = O.common(files, len(files))
    # a later file holds a code twice that one other file lacks: the count still reaches the number of files
    lacking = np.setdiff1d(np.intersect1d(files[0], files[5]), files[9])
    lacking = lacking[np.isin(lacking, O.common([f for j, f in enumerate(files) if j != 9], len(files) - 1))][:3]
    assert len(lacking) > 0
    f2, t2 = list(files), list(taxs)
    order = np.argsort(np.concatenate([f2[5], lacking]), kind="stable")
    f2[5] = np.concatenate([files[5], lacking])[order]
    t2[5] = np.concatenate([taxs[5], _taxids(lacking, T, 99)])[order]
    ok = both(f2, t2)
    assert np.isin(lacking, ok).all() and len(ok) == len(core) + len(lacking)
    # duplicates in the FIRST file collapse to one count (the last taxid stays)
    f2, t2 = list(files), list(taxs)
    f2[0] = np.repeat(files[0], 2)
    t2[0] = np.stack([taxs[0], _taxids(files[0], T, 98)], 1).reshape(-1)
    both(f2, t2)
    # an empty file: nothing reaches the count; an unsorted file: the map does not care
    f2, t2 = list(files), list(taxs)
    f2[3], t2[3] = np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    assert len(both(f2, t2)) == 0
    f2, t2 = list(files), list(taxs)
    perm = np.random.default_rng(5).permutation(len(f2[6]))
    f2[6], t2[6] = f2[6][perm], t2[6][perm]
    assert np.array_equal(both(f2, t2), core)
    # the all-ones code everywhere / in later files only
    ones = np.array([2**64 - 1], np.uint64)
    f2 = [np.concatenate([f, ones]) for f in files]
    t2 = [np.concatenate([t, np.array([7], np.uint32)]) for t in taxs]
    assert len(both(f2, t2)) == len(core) + 1
    both([files[0]] + f2[1:], [taxs[0]] + t2[1:])
    # some streams without taxids (taken as 0: the LCA rule's absorbing value)
    t2 = list(taxs)
    t2[2] = None
    gk, gt = ctx.common(files, len(files), t2)
    t3 = list(taxs)
    t3[2] = np.zeros(len(files[2]), np.uint32)
    ok, ot = O.common(files, len(files), t3, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_long_runs_fold_taxids_by_the_wave():
    """Runs of one code longer than eight records (merged chunk files, `common` over many files) have their taxids folded
    by the whole wave through pre-order numbers instead of by the head's lane; the reference folds LCA(LCA(t0, t1), t2) ...
    (sort.go:491, common.go:265).  Same answers as the oracle for run lengths around every boundary (1, 8, 9, 10, 63..66,
    72, 73, 200, runs across tile boundaries, one run of 9000) and the awkward ids: a forest, merged ids, 0, unknown ids,
    every record the same old / unknown id."""
    from oracle import oracle as O
    from unikmer_amd import lib as L
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
    lens = np.concatenate([rng.choice([1, 2, 8, 9, 10, 11, 63, 64, 65, 66, 72, 73, 74, 137, 200, 1000], 1500), [9000, 1, 5000]])
    codes = np.cumsum(rng.integers(1, 1 << 30, len(lens)).astype(np.uint64))
    keys = np.repeat(codes, lens)
    theme = np.repeat(rng.integers(0, 5, len(lens)), lens)
    fixed = np.repeat(rng.choice(pool, len(lens)), lens)
    t = rng.choice(pool, len(keys))
    t[theme == 0] = fixed[theme == 0]                                               # every record the same id
    t[theme == 1] = rng.integers(100, 160, int((theme == 1).sum()))                 # one chain
    t[theme == 2] = rng.integers(5000, 5121, int((theme == 2).sum()))               # one tree
    late = (theme == 3) & (rng.random(len(keys)) < 0.98)                            # the same id but for a few records
    t[late] = fixed[late]
    t = t.astype(np.uint32)
    for mode in (L.UNIQUE, L.REPEATED, L.SINGLETON, L.REPEATED_CHUNK):
        gk, gt = c.unique(keys, t, mode)
        ok, ot = O.unique(keys, t, mode, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), mode
    # the same sequence as a `common` over files: record i of a run goes to file i (every file strictly increasing)
    rank = np.arange(len(keys)) - np.repeat(np.cumsum(lens) - lens, lens)
    nfiles = 64
    short = rank < nfiles  # (longer runs: their first 64 records)
    files = [keys[short & (rank == f)] for f in range(nfiles)]
    taxs = [t[short & (rank == f)] for f in range(nfiles)]
    for thr in (1, 2, 9, 10, 11, 63, 64):
        gk, gt = c.common(files, thr, taxs)
        ok, ot = O.common(files, thr, taxs, tax)
        assert len(ok) > 0 and np.array_equal(gk, ok) and np.array_equal(gt, ot), thr
        assert np.array_equal(c.common(files, thr), O.common(files, thr))
    c.close()


def test_stream_table_is_reusable_across_operations(env):
    """Context.stream_table: the pointer / length tables of a file set built once and passed to several n-way calls (the
    C ABI's own arguments; UKM_F_DEVICE_STREAMS is set for device tensors) -- same results as the Python lists"""
    import torch
    O, L, ctx, tax, T = env
    U = _universe(20_000)
    files = [U[_member(len(U), f, 0.8, 3)] for f in range(70)]
    taxs = [_taxids(f, T, i) for i, f in enumerate(files)]
    dfiles = [torch.from_numpy(f.view(np.int64)).cuda() for f in files]
    dtaxs = [torch.from_numpy(t.view(np.int32)).cuda() for t in taxs]
    tab = ctx.stream_table(dfiles, dtaxs)
    dn = lambda r: (r[0].cpu().numpy().view(np.uint64), r[1].cpu().numpy().view(np.uint32))
    for fn, ofn in ((ctx.inter, O.inter), (ctx.diff, O.diff), (ctx.union, O.union)):
        gk, gt = dn(fn(tab))
        ok, ot = ofn(files, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), fn
    gk, gt = dn(ctx.common(tab, 60))
    ok, ot = O.common(files, 60, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = dn(ctx.merge_k(tab, mode=L.REPEATED))
    ok, ot = O.merge_k(files, taxs, mode=O.REPEATED, tax=tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # host arrays through the same object (no device-streams flag: they are staged)
    tabh = ctx.stream_table(files, taxs)
    gk, gt = ctx.inter(tabh)
    ok, ot = O.inter(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_merge_top_level_of_two_children_through_the_tile_kernel(env, monkeypatch):
    """A keep-everything merge of >= 2^20 records whose last level has two children (9..16 and 65..128 streams at fan-in 8)
    runs that level as a 2-way merge through the set-op tile kernel: the result is still the stable sort of the
    concatenation (util-sort.go:196-225: equal codes in stream order), with duplicates inside the streams, for plain keys
    and with taxids, and the same as with the k-way kernel at the top (UKM_KWAY_TOP2=0)."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(41)
    for nstreams, per in ((12, 110_000), (100, 14_000), (9, 150_000)):
        streams = [np.sort(rng.integers(0, 1 << 22, per + 977 * i).astype(np.uint64)) for i in range(nstreams)]   # many ties
        taxs = [_taxids(s + np.uint64(i), T, i) for i, s in enumerate(streams)]
        cat, tcat = np.concatenate(streams), np.concatenate(taxs)
        assert len(cat) >= 1 << 20
        o = np.argsort(cat, kind="stable")
        for knob in (None, "0"):
            if knob is None:
                monkeypatch.delenv("UKM_KWAY_TOP2", raising=False)
            else:
                monkeypatch.setenv("UKM_KWAY_TOP2", knob)
            gk, gt = ctx.merge_k(streams, taxs, mode=L.PLAIN)
            assert np.array_equal(gk, cat[o]) and np.array_equal(gt, tcat[o]), (nstreams, knob)
            assert np.array_equal(ctx.merge_k(streams, mode=L.PLAIN), cat[o])
            gk, gt = ctx.merge_k(streams, taxs, mode=L.REPEATED)
            ok, ot = O.merge_k(streams, taxs, mode=O.REPEATED, tax=tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (nstreams, knob)
        monkeypatch.delenv("UKM_KWAY_TOP2", raising=False)
    # an unsorted stream: the call still answers (concatenate + sort route)
    streams[3] = streams[3][::-1].copy()
    assert np.array_equal(ctx.merge_k(streams, mode=L.PLAIN), np.sort(np.concatenate(streams)))
