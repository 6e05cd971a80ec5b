"""Seeded randomised parity sweep on the GPU: many small / medium shapes around the tile geometry
(tile = 9728 merged items, 6656 with taxids or ranks), with duplicates, empty sides, taxids on
one or both sides and all three operations, each compared bit-exactly with the CPU oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import synth_tree  # noqa: E402

pytestmark = pytest.mark.gpu

# UKM_STRESS_SEEDS=N widens every sweep below to N seeds (default: the small fixed sets used in CI)
_EXTRA = int(os.environ.get("UKM_STRESS_SEEDS", "0"))


def _seeds(n):
    return range(max(n, _EXTRA))


@pytest.fixture(scope="module")
def env():
    from oracle import oracle as O
    from unikmer_amd import lib as L
    ctx = L.Context(0)
    child, parent = synth_tree(5, 4)
    ctx.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    return O, L, ctx, tax, len(child)


def _draw(rng, n, universe, dup_rate):
    if n == 0:
        return np.empty(0, np.uint64)
    a = np.sort(rng.choice(universe, size=n, replace=False)) if n <= len(universe) else np.sort(rng.choice(universe, size=n))
    if dup_rate > 0 and n > 1:
        m = rng.random(n) < dup_rate
        a[1:][m[1:]] = a[:-1][m[1:]]          # copy the left neighbour: runs of equal codes
        a = np.sort(a)
    return a.astype(np.uint64)


SIZES = [0, 1, 2, 63, 64, 65, 511, 512, 6143, 6144, 6145, 6655, 6656, 6657, 9727, 9728, 9729, 13312, 19456, 30000, 100_000]


@pytest.mark.parametrize("seed", _seeds(6))
def test_random_setops_plain_and_multiset(env, seed):
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(1000 + seed)
    for it in range(40):
        na, nb = int(rng.choice(SIZES)), int(rng.choice(SIZES))
        span = max(4, int((na + nb) * rng.choice([0.6, 1.0, 3.0, 50.0])))
        universe = np.sort(rng.choice(1 << 40, size=span, replace=False)).astype(np.uint64) if span < (1 << 20) else \
            rng.integers(0, 1 << 62, span, dtype=np.uint64)
        universe = np.unique(universe)
        dup = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
        a = _draw(rng, min(na, len(universe)), universe, dup)
        b = _draw(rng, min(nb, len(universe)), universe, dup)
        for op, ref in ((L.OP_UNION, O.union), (L.OP_INTER, O.inter), (L.OP_DIFF, O.diff)):
            if op == L.OP_INTER and len(b) == 0 and len(a):
                continue                       # the reference keeps the running result (quirk covered elsewhere)
            got = ctx.setop2(op, a, b)
            exp = ref([a, b])
            exp = np.sort(exp) if op == L.OP_UNION else exp
            assert np.array_equal(got, exp), (seed, it, op, na, nb, dup)


@pytest.mark.parametrize("seed", _seeds(4))
def test_random_setops_with_taxids(env, seed):
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(2000 + seed)
    for it in range(25):
        na, nb = int(rng.choice(SIZES[:-1])), int(rng.choice(SIZES[:-1]))
        universe = np.unique(rng.integers(0, 1 << 44, max(8, int((na + nb) * rng.choice([0.7, 2.0]))), dtype=np.uint64))
        a = _draw(rng, min(na, len(universe)), universe, 0.0)
        b = _draw(rng, min(nb, len(universe)), universe, 0.0)
        ta = rng.integers(0, T + 1, len(a)).astype(np.uint32)       # includes taxid 0
        tb = rng.integers(0, T + 1, len(b)).astype(np.uint32)
        which = int(rng.integers(0, 3))                              # both / only A / only B carry taxids
        xa, xb = (ta, tb) if which == 0 else ((ta, None) if which == 1 else (None, tb))
        # union / inter (mix-taxid when one side has none)
        gk, gt = ctx.setop2(L.OP_UNION, a, b, xa, xb)
        za = xa if xa is not None else np.zeros(len(a), np.uint32)   # a stream without taxids counts as taxid 0
        zb = xb if xb is not None else np.zeros(len(b), np.uint32)
        ek, et = O.union([a, b], [za, zb], tax)
        o = np.argsort(ek, kind="stable")
        assert np.array_equal(gk, ek[o]) and np.array_equal(gt, et[o]), (seed, it, "union", which)
        if len(b) or not len(a):
            mix = which != 0
            gk, gt = ctx.setop2(L.OP_INTER, a, b, xa, xb, flags=L.F_MIX_TAXID if mix else 0)
            ek, et = O.inter([a, b], [xa, xb], tax, mix_taxid=mix)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "inter", which)
        if which == 0:
            gk, gt = ctx.setop2(L.OP_DIFF, a, b, ta, tb, flags=L.F_CMP_TAXID)
            ek, et = O.diff([a, b], [ta, tb], tax, compare_taxid=True)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "diff -t")


@pytest.mark.parametrize("seed", _seeds(3))
def test_random_sort_scan_nway(env, seed):
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(3000 + seed)
    for it in range(12):
        n = int(rng.choice([0, 1, 2, 100, 7167, 7168, 7169, 11264, 11265, 50_000, 200_000]))
        bits = int(rng.choice([8, 20, 42, 62, 64]))
        x = rng.integers(0, 1 << bits, n, dtype=np.uint64) if bits < 64 else rng.integers(0, 1 << 64, n, dtype=np.uint64)
        assert np.array_equal(ctx.sort_u64(x.copy(), bits), np.sort(x)), (seed, it, n, bits)
        v = rng.integers(0, T + 1, n).astype(np.uint32)
        k2, v2 = ctx.sort_pairs(x.copy(), v.copy(), bits)
        o = np.argsort(x, kind="stable")
        assert np.array_equal(k2, x[o]) and np.array_equal(v2, v[o]), (seed, it, "pairs")
        srt = x[o]
        for mode in (L.UNIQUE, L.REPEATED, L.SINGLETON, L.REPEATED_CHUNK, L.PLAIN):
            assert np.array_equal(ctx.unique(srt, mode=mode), O.unique(srt, mode=mode)), (seed, it, "unique", mode)
        if n:
            gk, gt = ctx.unique(srt, v2, mode=L.UNIQUE)
            ek, et = O.unique(srt, v2, mode=O.UNIQUE, tax=tax)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "unique+lca")
    for it in range(6):
        nf = int(rng.integers(1, 9))
        files = []
        for f in range(nf):
            m = int(rng.choice([0, 1, 500, 9728, 20_000]))
            files.append(np.unique(rng.integers(0, 60_000, m).astype(np.uint64)))
        assert np.array_equal(ctx.union(files), np.sort(O.union(files))), (seed, it, "union", nf)
        assert np.array_equal(ctx.merge_k(files, mode=L.PLAIN), O.merge_k(files, mode=O.PLAIN)), (seed, it, "merge")
        assert np.array_equal(ctx.merge_k(files, mode=L.REPEATED), O.merge_k(files, mode=O.REPEATED))
        assert np.array_equal(ctx.merge_k(files, mode=L.REPEATED, final_round=False), O.merge_k(files, mode=O.REPEATED, final_round=False))
        thr = int(rng.integers(1, nf + 1))
        assert np.array_equal(ctx.common(files, thr), O.common(files, thr)), (seed, it, "common", thr)
        if len(files[0]):
            assert np.array_equal(ctx.diff(files), O.diff(files)), (seed, it, "diff")
        if all(len(f) for f in files):
            assert np.array_equal(ctx.inter(files), O.inter(files)), (seed, it, "inter")


@pytest.mark.parametrize("seed", _seeds(3))
def test_random_windows(env, seed):
    """encode / ntHash / Scaled filter / minimizer over random ragged record layouts (empty records, records
    shorter than k, records across tile borders of 2048 windows, IUPAC and lower-case bases, circular)."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(4000 + seed)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNRYKMn", dtype=np.uint8)
    for it in range(10):
        nrec = int(rng.choice([1, 2, 7, 40, 300]))
        lens = rng.choice([0, 1, 5, 31, 32, 150, 151, 2047, 2048, 2049, 5000], size=nrec)
        cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        bases = alphabet[rng.integers(0, len(alphabet), int(cuts[-1]))]
        k = int(rng.choice([1, 5, 21, 31, 32]))
        circ = bool(rng.integers(0, 2))
        canon = bool(rng.integers(0, 2))
        if cuts[-1] == 0:
            continue
        assert np.array_equal(ctx.encode_kmers(bases, cuts, k, canonical=canon, circular=circ),
                              O.count_windows(bases, cuts, k, canonical=canon, circular=circ)), (seed, it, "encode", k, circ)
        kh = int(rng.choice([1, 16, 31, 51, 64]))
        assert np.array_equal(ctx.nthash(bases, cuts, kh, canonical=canon, circular=circ),
                              O.count_windows(bases, cuts, kh, hashed=True, canonical=canon, circular=circ)), (seed, it, "nthash", kh)
        mh = O.max_hash(int(rng.choice([2, 7, 100])))
        assert np.array_equal(ctx.nthash(bases, cuts, kh, canonical=True, circular=circ, max_hash=mh),
                              O.count_windows(bases, cuts, kh, hashed=True, canonical=True, circular=circ, max_hash=mh)), (seed, it, "scaled")
        w = int(rng.choice([1, 3, 15, 100]))
        hs, ps = [], []
        for r in range(nrec):
            seq = bases[int(cuts[r]):int(cuts[r + 1])]
            try:
                h, p = O.minimizer(seq, kh, w, circular=circ)
            except ValueError:
                continue
            hs.append(h); ps.append(p)
        eh = np.concatenate(hs) if hs else np.empty(0, np.uint64)
        ep = np.concatenate(ps) if ps else np.empty(0, np.uint64)
        gh, gp = ctx.minimizer(bases, cuts, kh, w, circular=circ, with_pos=True)
        assert np.array_equal(gh, eh) and np.array_equal(gp, ep), (seed, it, "minimizer", kh, w, circ)


@pytest.mark.parametrize("seed", _seeds(4))
def test_random_windows_strip_kernels(env, seed, monkeypatch):
    """The rolling strip kernels forced on (UKM_WIN_STRIP=1: every window; UKM_NTHASH_STRIP=1: Scaled sketch) over
    random layouts: a few long records mixed with empty / short / k-sized ones, cuts next to row (16) and strip
    boundaries, random strip lengths, IUPAC / lower-case / N runs, and sizes from a fraction of a tile to several."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(9000 + seed)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTacgtNRYKMn", dtype=np.uint8)
    monkeypatch.setenv("UKM_WIN_STRIP", "1")
    monkeypatch.setenv("UKM_NTHASH_STRIP", "1")
    for it in range(6):
        n = int(rng.choice([700, 16_384, 70_000, 300_000, 1_100_000]))
        ncut = int(rng.choice([0, 1, 3, 12, 60]))
        inner = rng.integers(0, n + 1, ncut)
        near = []
        for c in inner[: 6]:   # cuts that leave records of 0, 1, k-1, k, 15..17 bases
            near += [int(c) + d for d in (0, 1, 15, 16, 17, 30, 31, 32) if int(c) + d <= n]
        cuts = np.unique(np.concatenate([[0, n], inner, near])).astype(np.uint64)
        if rng.integers(0, 2):
            cuts = np.sort(np.concatenate([cuts, cuts[rng.integers(0, len(cuts), 3)]]))   # empty records
        bases = alphabet[rng.integers(0, len(alphabet), n)]
        if rng.integers(0, 2):
            a = int(rng.integers(0, n))
            bases[a:a + int(rng.integers(1, 200))] = ord("N")
        Ls = str(int(rng.choice([64, 128, 192, 256])))
        monkeypatch.setenv("UKM_WIN_STRIP_L", Ls)
        k = int(rng.choice([1, 3, 15, 16, 17, 21, 31, 32]))
        canon = bool(rng.integers(0, 2))
        assert np.array_equal(ctx.encode_kmers(bases, cuts, k, canonical=canon),
                              O.count_windows(bases, cuts, k, canonical=canon)), (seed, it, "codes", k, canon, Ls, n)
        kh = int(rng.choice([1, 2, 16, 31, 32, 33, 51, 63, 64]))
        assert np.array_equal(ctx.nthash(bases, cuts, kh, canonical=canon),
                              O.count_windows(bases, cuts, kh, hashed=True, canonical=canon)), (seed, it, "nthash", kh, canon, Ls, n)
        mh = O.max_hash(int(rng.choice([50, 300, 2000])))
        assert np.array_equal(ctx.nthash(bases, cuts, kh, canonical=canon, max_hash=mh),
                              O.count_windows(bases, cuts, kh, hashed=True, canonical=canon, max_hash=mh)), (seed, it, "scaled", kh)
        w = int(rng.choice([3, 15]))
        hs, ps = [], []
        for r in range(len(cuts) - 1):
            seq = bases[int(cuts[r]):int(cuts[r + 1])]
            try:
                h, p = O.minimizer(seq, kh, w)
            except ValueError:
                continue
            hs.append(h); ps.append(p)
        eh = np.concatenate(hs) if hs else np.empty(0, np.uint64)
        ep = np.concatenate(ps) if ps else np.empty(0, np.uint64)
        gh, gp = ctx.minimizer(bases, cuts, kh, w, with_pos=True)
        assert np.array_equal(gh, eh) and np.array_equal(gp, ep), (seed, it, "minimizer", kh, w)


@pytest.mark.parametrize("seed", _seeds(4))
def test_random_nway_with_taxids(env, seed):
    """n-way union / inter (-m) / diff (-t) / common / merge (-u, -d, chunk rounds) with per-record taxids,
    some files without taxids (mix), taxid 0 records, empty files in the middle."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(5000 + seed)
    for it in range(10):
        nf = int(rng.integers(2, 7))
        files, taxs = [], []
        for f in range(nf):
            m = int(rng.choice([0, 3, 700, 6144, 6145, 15_000])) if f else int(rng.choice([700, 6144, 15_000]))
            k = np.unique(rng.integers(0, 40_000, m).astype(np.uint64))
            files.append(k)
            taxs.append(rng.integers(0, T + 1, len(k)).astype(np.uint32))
        # all files with taxids
        gk, gt = ctx.union(files, taxs)
        ek, et = O.union(files, taxs, tax)
        o = np.argsort(ek, kind="stable")
        assert np.array_equal(gk, ek[o]) and np.array_equal(gt, et[o]), (seed, it, "union")
        for mode, om in ((L.UNIQUE, O.UNIQUE), (L.REPEATED, O.REPEATED), (L.PLAIN, O.PLAIN)):
            for fr in (True, False):
                gk, gt = ctx.merge_k(files, taxs, mode=mode, final_round=fr)
                ek, et = O.merge_k(files, taxs, mode=om, final_round=fr, tax=tax)
                assert np.array_equal(gk, ek), (seed, it, "merge keys", mode, fr)
                if mode != L.PLAIN:   # plain keeps every record: the order of equal codes' taxids is the stream order
                    assert np.array_equal(gt, et), (seed, it, "merge taxids", mode, fr)
                else:
                    assert np.array_equal(np.sort(gt), np.sort(et))
        thr = int(rng.integers(1, nf + 1))
        gk, gt = ctx.common(files, thr, taxs)
        ek, et = O.common(files, thr, taxs, tax)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "common", thr)
        gk, gt = ctx.diff(files, taxs, compare_taxid=True)
        ek, et = O.diff(files, taxs, tax, compare_taxid=True)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "diff -t")
        gk, gt = ctx.diff(files, taxs)
        ek, et = O.diff(files, taxs, tax)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "diff")
        if all(len(f) for f in files):
            gk, gt = ctx.inter(files, taxs)
            ek, et = O.inter(files, taxs, tax)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "inter")
            # mix: drop the taxids of one later file
            drop = int(rng.integers(1, nf))
            mixed = [t if i != drop else None for i, t in enumerate(taxs)]
            gk, gt = ctx.inter(files, mixed, mix_taxid=True)
            ek, et = O.inter(files, mixed, tax, mix_taxid=True)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (seed, it, "inter -m", drop)


@pytest.mark.parametrize("seed", _seeds(3))
def test_strip_windows_on_ragged_reads(env, seed, monkeypatch):
    """Reads through the strip kernel: every lane shifts its strip for the record it starts in, so with records of
    80-260 bases every lane has its own shift, shifts reach back into the previous record, most strips cross a
    record boundary and rows are cut at both ends — codes and ntHash against the oracle for k on both sides of 16 /
    32, both strip lengths, with reads shorter than k and empty ones in between."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(9100 + seed)
    monkeypatch.setenv("UKM_WIN_STRIP", "1")
    n_target = 1_500_000
    lens = rng.integers(80, 261, n_target // 170)
    lens[rng.integers(0, len(lens), len(lens) // 40)] = rng.integers(0, 34, len(lens) // 40)   # shorter than k, empty
    if seed % 2:
        lens[:] = 150   # fixed-length reads: every record's gap step is k - 1
    cuts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(cuts[-1])
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    for Ls in ("64", "128"):
        monkeypatch.setenv("UKM_WIN_STRIP_L", Ls)
        for k in (9, 16, 17, 30, 31, 32):
            canon = bool(rng.integers(0, 2))
            assert np.array_equal(ctx.encode_kmers(bases, cuts, k, canonical=canon),
                                  O.count_windows(bases, cuts, k, canonical=canon)), (seed, "codes", k, canon, Ls)
        for kh in (15, 33, 50, 51, 64):
            canon = bool(rng.integers(0, 2))
            assert np.array_equal(ctx.nthash(bases, cuts, kh, canonical=canon),
                                  O.count_windows(bases, cuts, kh, hashed=True, canonical=canon)), (seed, "nthash", kh, canon, Ls)
    # an illegal base in a read that is emitted / in one that is shorter than k
    monkeypatch.delenv("UKM_WIN_STRIP_L", raising=False)
    long_r = int(np.argmax(lens >= 80))
    bad = bases.copy()
    bad[int(cuts[long_r]) + 40] = ord("*")
    with pytest.raises(L.IllegalBaseError):
        ctx.encode_kmers(bad, cuts, 31)
    short = np.flatnonzero((lens > 0) & (lens < 31))
    if len(short):
        bad = bases.copy()
        bad[int(cuts[short[0]])] = ord("*")
        assert np.array_equal(ctx.encode_kmers(bad, cuts, 31), O.count_windows(bad, cuts, 31))


@pytest.mark.parametrize("seed", _seeds(3))
def test_random_probe_paths(env, seed, monkeypatch):
    """The hash-probe routes forced on (UKM_PUNION=2: union without its size and hit-rate guards; the probe fold takes
    inter / diff / diff -t from four streams on, UKM_PFOLD_TAX=1 also inter with taxids) over random shapes: 9-40
    streams of 0 - 30k codes with every degree of overlap, value spaces from a few hundred codes (dense duplicates across
    streams) to 62 bits, taxids on all / some / no streams, against the oracle's sequential folds."""
    O, L, ctx, tax, T = env
    rng = np.random.default_rng(9300 + seed)
    monkeypatch.setenv("UKM_PUNION", "2")
    monkeypatch.setenv("UKM_PFOLD_TAX", "1")
    for it in range(8):
        nf = int(rng.integers(9, 41))
        space = int(rng.choice([300, 5_000, 200_000, 1 << 40, 1 << 62]))
        base = np.unique(rng.integers(0, space, int(rng.choice([50, 3_000, 30_000])), dtype=np.uint64))
        files = []
        for f in range(nf):
            kind = rng.integers(0, 4)
            if kind == 0:       # a random subset of a common pool
                x = base[rng.random(len(base)) < rng.random()]
            elif kind == 1:     # private codes
                x = np.unique(rng.integers(0, space, int(rng.integers(0, 4_000)), dtype=np.uint64))
            elif kind == 2:     # the pool plus private codes
                x = np.unique(np.concatenate([base[rng.random(len(base)) < 0.8],
                                              rng.integers(0, space, int(rng.integers(0, 500)), dtype=np.uint64)]))
            else:               # almost the whole pool
                x = base[rng.random(len(base)) < 0.97]
            files.append(x)
        taxs = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in files]
        assert np.array_equal(ctx.union(files), O.union(files)), (seed, it, "union", nf, space)
        gk, gt = ctx.union(files, taxs)
        ok, ot = O.union(files, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (seed, it, "union+tax", nf, space)
        # `common` through the counting probes (guards off): any threshold, files with duplicates inside, empty files
        dup = [np.sort(np.concatenate([f, f[::3]])) if (i % 4 == 1 and len(f)) else f for i, f in enumerate(files)]
        for thr in {1, 2, int(rng.integers(1, nf + 2)), nf}:
            assert np.array_equal(ctx.common(dup, thr), O.common(dup, thr)), (seed, it, "common", thr, nf, space)
            dtax = [(1 + rng.integers(0, T, len(f))).astype(np.uint32) for f in dup]
            gk, gt = ctx.common(dup, thr, dtax)
            ok, ot = O.common(dup, thr, dtax, tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (seed, it, "common+tax", thr, nf, space)
        live = [f for f in files if len(f)]
        ltax = [t for f, t in zip(files, taxs) if len(f)]
        if len(live) >= 4:
            assert np.array_equal(ctx.inter(live), O.inter(live)), (seed, it, "inter")
            gk, gt = ctx.inter(live, ltax)
            ok, ot = O.inter(live, ltax, tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (seed, it, "inter+tax")
        if len(files[0]):
            assert np.array_equal(ctx.diff(files), O.diff(files)), (seed, it, "diff")
            for cmp_t in (False, True):
                gk, gt = ctx.diff(files, taxs, compare_taxid=cmp_t)
                ok, ot = O.diff(files, taxs, tax, compare_taxid=cmp_t)
                assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (seed, it, "diff+tax", cmp_t)
            some = [t if i % 3 else None for i, t in enumerate(taxs)]   # streams without taxids among streams with
            some[0] = taxs[0]
            gk, gt = ctx.diff(files, some, compare_taxid=True)
            ok, ot = O.diff(files, some, tax, compare_taxid=True)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (seed, it, "diff+some tax")


@pytest.mark.parametrize("seed", _seeds(3))
def test_random_sorts_through_the_bucket_route(env, seed):
    """Sorts of 9e6 - 2.5e7 keys (the LDS bucket route of ukm_sort.hip) on seeded key distributions: even, power-law
    clusters (heavy buckets, sometimes the fallback), smooth skew like canonical k-mers (min of two draws), few distinct
    values, random key widths; keys only against torch.sort, with taxids against a stable torch.sort."""
    import torch
    O, L, _, tax, T = env
    dev = torch.device("cuda:0")
    # (device tensors made by torch: the context works on torch's stream, so that the sort is ordered behind them)
    ctx = L.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(9400 + seed)
    rng = np.random.default_rng(9400 + seed)

    def rnd(n, bits):
        hi = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
        lo = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
        x = (hi << 33) ^ (lo << 2) ^ torch.randint(0, 4, (n,), device=dev, generator=g, dtype=torch.int64)
        return x if bits == 64 else x & ((1 << bits) - 1)

    for it in range(4):
        n = int(rng.integers(9_000_000, 25_000_000))
        bits = int(rng.choice([40, 52, 62, 64]))
        kind = int(rng.integers(0, 4))
        x = rnd(n, bits)
        if kind == 1:      # power-law clusters: a random share of the keys gets its top bits from a small set
            share = float(rng.choice([0.02, 0.2, 0.6]))
            m = torch.rand(n, device=dev, generator=g) < share
            tops = torch.randint(0, 1 << 12, (int(rng.choice([3, 40, 2000])),), device=dev, generator=g, dtype=torch.int64)
            pick = tops[torch.randint(0, tops.numel(), (n,), device=dev, generator=g)]
            low = bits - 16 if bits < 64 else 48
            x = torch.where(m, (x & ((1 << low) - 1)) | (pick << low), x)
            if bits < 64:
                x &= (1 << bits) - 1
        elif kind == 2:    # canonical-like: the smaller of two draws (unsigned order)
            y = rnd(n, bits)
            f = -1 << 63
            x = torch.where((x ^ f) < (y ^ f), x, y)
        elif kind == 3:    # few distinct values
            vals = rnd(int(rng.choice([1, 17, 5000])), bits)
            x = vals[torch.randint(0, vals.numel(), (n,), device=dev, generator=g)]
        f = -1 << 63
        srt = torch.sort(x ^ f, stable=True)
        exp, order = srt.values ^ f, srt.indices
        w = x.clone()
        ctx.sort_u64(w, bits)
        assert torch.equal(w, exp), (seed, it, n, bits, kind, "keys")
        v = torch.arange(n, dtype=torch.int32, device=dev)
        w.copy_(x)
        ctx.sort_pairs(w, v, bits)
        assert torch.equal(w, exp) and torch.equal(v.long(), order), (seed, it, n, bits, kind, "pairs")
        del x, w, v, exp, order, srt
    torch.cuda.synchronize()
    ctx.close()


def test_bucket_route_narrows_keys_declared_64_bits_wide():
    """A caller that cannot narrow key_bits (hashes: 64) but whose keys happen to use 57 .. 63 bits: the bucket route
    narrows to the bits in use (it used to keep 64 for > 56 bits: a sparsely populated top digit, two wasted scatter
    passes, and the context marked as one that has met crowded keys).  Same result as torch.sort, keys and stable pairs,
    and the context does NOT start sampling afterwards (a second, even, sort still takes the route: same result)."""
    import torch
    from unikmer_amd import lib as L
    dev = torch.device("cuda:0")
    ctx = L.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    n = 10_000_000
    for bits in (57, 60, 63):
        hi = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
        lo = torch.randint(0, 1 << 32, (n,), device=dev, generator=g, dtype=torch.int64)
        x = ((hi << 32) ^ lo) & ((1 << bits) - 1)
        exp = torch.sort(x, stable=True)          # (< 2^63: signed order = unsigned order)
        w = x.clone()
        ctx.sort_u64(w, 64)
        assert torch.equal(w, exp.values), bits
        v = torch.arange(n, dtype=torch.int32, device=dev)
        w.copy_(x)
        ctx.sort_pairs(w, v, 64)
        assert torch.equal(w, exp.values) and torch.equal(v.long(), exp.indices), bits
    torch.cuda.synchronize()
    ctx.close()
