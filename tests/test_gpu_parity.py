"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.
Run on the MI355X box with `pytest -m gpu`."""
import os
import numpy as np
import pytest

from conftest import AMUC, IAI39, MG1655, splitmix64, synth_tree

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572  # "unikmer" 


@pytest.fixture(scope="module")
def ctx():
    from unikmer_amd import lib
    c = lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def L():
    from unikmer_amd import lib
    return lib


def synth_sets(n_universe, gap_bits, seed=SEED, probs=(1, 1, 2)):
    """universe = prefix sum of random gaps; membership by hash bits
    (0 -> A only, 1 -> B only, 2/3 -> both)."""
    j = np.arange(n_universe, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64(np.uint64(seed) ^ j) & np.uint64((1 << gap_bits) - 1))
    U = np.cumsum(gaps, dtype=np.uint64)
    m = splitmix64(np.uint64(seed + 1) ^ j) & np.uint64(3)
    A = U[(m == 0) | (m >= 2)]
    B = U[(m == 1) | (m >= 2)]
    return A, B


def taxids_for(codes, T, seed=SEED + 2):
    return (np.uint64(1) + splitmix64(np.uint64(seed) ^ codes) % np.uint64(T)).astype(np.uint32)


@pytest.fixture(scope="module")
def tree(ctx, O):
    child, parent = synth_tree(depth=5, arity=8)
    ctx.taxonomy_load(child, parent)
    return O.Taxonomy(child, parent), len(child)


# ---------------------------------------------------------------------------------- set ops
@pytest.mark.parametrize("n", [0, 1, 5, 4095, 4096, 4097, 100_000, 1_333_000, 3_100_000])
def test_setop2_matches_oracle(ctx, O, L, n):
    A, B = synth_sets(n, 22) if n else (np.empty(0, np.uint64), np.empty(0, np.uint64))
    u = ctx.setop2(L.OP_UNION, A, B)
    i = ctx.setop2(L.OP_INTER, A, B)
    d = ctx.setop2(L.OP_DIFF, A, B)
    assert np.array_equal(u, O.union([A, B]))
    assert np.array_equal(i, O.inter([A, B]) if len(A) else np.empty(0, np.uint64))
    assert np.array_equal(d, O.diff([A, B]))
    # size-independent identities
    assert len(u) == len(A) + len(B) - len(i)
    assert len(d) == len(A) - len(i)


def test_setop2_edge_cases(ctx, O, L):
    e = np.empty(0, np.uint64)
    a = np.array([0, 1, 2, 2**62 - 1, 2**64 - 1], dtype=np.uint64)
    b = np.array([0, 2, 3, 2**64 - 1], dtype=np.uint64)
    for x, y in [(a, b), (b, a), (a, a), (a, e), (e, a), (a[:1], a[:1]), (a[:1], b[1:2])]:
        assert np.array_equal(ctx.setop2(L.OP_UNION, x, y), np.union1d(x, y))
        assert np.array_equal(ctx.setop2(L.OP_INTER, x, y), np.intersect1d(x, y))
        assert np.array_equal(ctx.setop2(L.OP_DIFF, x, y), np.setdiff1d(x, y))
    # disjoint / interleaved / all-equal large
    x = np.arange(0, 200000, 2, dtype=np.uint64)
    y = np.arange(1, 200001, 2, dtype=np.uint64)
    assert np.array_equal(ctx.setop2(L.OP_UNION, x, y), np.arange(200000, dtype=np.uint64))
    assert len(ctx.setop2(L.OP_INTER, x, y)) == 0
    assert np.array_equal(ctx.setop2(L.OP_DIFF, x, y), x)
    assert np.array_equal(ctx.setop2(L.OP_INTER, x, x), x)
    assert len(ctx.setop2(L.OP_DIFF, x, x)) == 0
    # all of A below all of B
    assert np.array_equal(ctx.setop2(L.OP_UNION, x, x + np.uint64(10**6)), np.concatenate([x, x + np.uint64(10**6)]))


def test_setop2_unsorted_is_an_error(ctx, L):
    a = np.array([5, 3, 9], dtype=np.uint64)
    b = np.array([1, 2, 3], dtype=np.uint64)
    with pytest.raises(L.UnsortedError):
        ctx.setop2(L.OP_INTER, a, b)


def test_setop2_capacity_error(ctx, L):
    A, B = synth_sets(10000, 20)
    out = np.empty(10, dtype=np.uint64)
    with pytest.raises(L.CapacityError):
        ctx.setop2(L.OP_UNION, A, B, out=out)


def test_setop2_multiset_semantics(ctx, O, L):
    # duplicates inside an input: the reference's 2-pointer advances both cursors on equality
    rng = np.random.default_rng(7)
    a = np.sort(rng.integers(0, 3000, 20000).astype(np.uint64))
    b = np.sort(rng.integers(0, 3000, 15000).astype(np.uint64))
    assert np.array_equal(ctx.setop2(L.OP_INTER, a, b), O.inter([a, b]))
    assert np.array_equal(ctx.setop2(L.OP_DIFF, a, b), O.diff([a, b]))
    assert np.array_equal(ctx.setop2(L.OP_UNION, a, b), O.union([a, b]))
    # long runs (exercise the lower_bound branch of rank-in-run)
    a = np.sort(np.concatenate([np.full(500, 7), np.full(100, 9), np.arange(100)]).astype(np.uint64))
    b = np.sort(np.concatenate([np.full(200, 7), np.full(300, 9), np.arange(50, 150)]).astype(np.uint64))
    assert np.array_equal(ctx.setop2(L.OP_INTER, a, b), O.inter([a, b]))
    assert np.array_equal(ctx.setop2(L.OP_DIFF, a, b), O.diff([a, b]))


@pytest.mark.parametrize("n", [10, 5000, 300_000])
def test_setop2_taxids_lca(ctx, O, L, tree, n):
    tax, T = tree
    A, B = synth_sets(n, 22)
    ta, tb = taxids_for(A, T), taxids_for(B, T, SEED + 5)
    uk, ut = ctx.setop2(L.OP_UNION, A, B, ta, tb)
    ok, ot = O.union([A, B], [ta, tb], tax)
    assert np.array_equal(uk, ok) and np.array_equal(ut, ot)
    ik, it = ctx.setop2(L.OP_INTER, A, B, ta, tb)
    ok, ot = O.inter([A, B], [ta, tb], tax)
    assert np.array_equal(ik, ok) and np.array_equal(it, ot)
    dk, dt = ctx.setop2(L.OP_DIFF, A, B, ta, tb, flags=L.F_CMP_TAXID)
    ok, ot = O.diff([A, B], [ta, tb], tax, compare_taxid=True)
    assert np.array_equal(dk, ok) and np.array_equal(dt, ot)
    dk, dt = ctx.setop2(L.OP_DIFF, A, B, ta, tb)
    ok, ot = O.diff([A, B], [ta, tb], tax)
    assert np.array_equal(dk, ok) and np.array_equal(dt, ot)
    # mix-taxid: zeros on either side
    ta0 = ta.copy(); ta0[::3] = 0
    tb0 = tb.copy(); tb0[::5] = 0
    ik, it = ctx.setop2(L.OP_INTER, A, B, ta0, tb0, flags=L.F_MIX_TAXID)
    ok, ot = O.inter([A, B], [ta0, tb0], tax, mix_taxid=True)
    assert np.array_equal(ik, ok) and np.array_equal(it, ot)


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_setop2_per_record_taxids_source_words(ctx, O, L, tree, monkeypatch, mode):
    """Round 5: `inter` with per-record taxids runs the plain-key kernel, whose epilogue writes one SOURCE WORD per output
    record, and a second launch turns the words into taxids (ukm_setops.hip: tile_flush_src / setop_taxid_gather_kernel).
    UKM_SETOP_SRC: 0 = the taxid instantiation everywhere (round 6: the default on a taxonomy with one-byte clade codes),
    1 = inter through source words, 2 = every operation that allows it.
    All three against the oracle on sizes around tile boundaries (a tile is 9728 merged records), on one-sided taxids
    (a file taxid on the other stream), zeros (mix-taxid), a match at a tile's last place, and empty inputs."""
    tax, T = tree
    monkeypatch.setenv("UKM_SETOP_SRC", mode)
    rng = np.random.default_rng(5)
    for n in (1, 9727, 9728, 9729, 40_000, 250_000):
        A, B = synth_sets(n, 20)
        ta, tb = taxids_for(A, T), taxids_for(B, T, SEED + 5)
        ta[::7] = 0
        tb[::5] = 0
        for op, ofn, kw in ((L.OP_UNION, O.union, {}), (L.OP_INTER, O.inter, {}), (L.OP_DIFF, O.diff, {})):
            gk, gt = ctx.setop2(op, A, B, ta, tb)
            ek, et = ofn([A, B], [ta, tb], tax, **kw)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, n, op)
        gk, gt = ctx.setop2(L.OP_INTER, A, B, ta, tb, flags=L.F_MIX_TAXID)
        ek, et = O.inter([A, B], [ta, tb], tax, mix_taxid=True)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, n, "mix")
        # per-record taxids on ONE stream, a file taxid on the other
        ft = int(rng.integers(1, T))
        gk, gt = ctx.setop2(L.OP_INTER, A, B, ta, ft)
        ek, et = O.inter([A, B], [ta, np.full(len(B), ft, np.uint32)], tax)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, n, "one-sided")
    # identical sets: every record a match, also the one at each tile's last place; and an empty side
    A = np.arange(1, 30001, dtype=np.uint64) * 3
    ta, tb = taxids_for(A, T), taxids_for(A, T, SEED + 9)
    gk, gt = ctx.setop2(L.OP_INTER, A, A, ta, tb)
    ek, et = O.inter([A, A], [ta, tb], tax)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    # (one more record in front of A: every tile's last place now holds the A half of a pair whose B half opens the next tile)
    A1 = np.concatenate([np.array([1], np.uint64), A])
    ta1 = taxids_for(A1, T)
    for op, ofn in ((L.OP_INTER, O.inter), (L.OP_UNION, O.union)):
        gk, gt = ctx.setop2(op, A1, A, ta1, tb)
        ek, et = ofn([A1, A], [ta1, tb], tax)
        assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, op, "pair across a tile boundary")
    e = np.empty(0, np.uint64)
    gk, gt = ctx.setop2(L.OP_INTER, A, e, ta, np.empty(0, np.uint32))
    assert len(gk) == 0 and len(gt) == 0


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_setop2_taxids_deep_forest_clade_paths(O, L, monkeypatch, mode):
    """Round 6: the clade-pair step reads two 16-level root paths out of LDS (ukm_device.h: lca_clade_pair_lds) and the
    LCAs of a tile are walked densely behind the merge loop (tile_merge_loop_deferred / tile_lca_dense).  A forest whose
    clade cut runs 40 levels down a spine (every spine node has two small subtrees beside the next one: the cut follows
    the largest subtree), so that pairs below DIFFERENT clade nodes share all 16 levels (the pair-table fallback), pairs
    in other trees (LCA 0), relatives inside one clade (root paths), merged ids, zero and unknown ids; union / inter /
    diff [-t] / inter --mix-taxid against the oracle's ancestor walk through all three routes (UKM_SETOP_SRC) and with the
    dense pass switched off."""
    monkeypatch.setenv("UKM_SETOP_SRC", mode)
    child, parent = [1], [1]
    nxt = 2
    spine = 1
    for d in range(40):                       # spine + two bushes of 1 + 3 + 9 nodes per level
        nodes = []
        for b in range(2):
            root = nxt; nxt += 1
            child.append(root); parent.append(spine); nodes.append(root)
            for k in range(3):
                m = nxt; nxt += 1
                child.append(m); parent.append(root)
                for k2 in range(3):
                    child.append(nxt); parent.append(m); nxt += 1
        s2 = nxt; nxt += 1
        child.append(s2); parent.append(spine)
        spine = s2
    for k in range(3000):                     # a large bush at the end of the spine: the cut goes all the way down
        child.append(nxt); parent.append(spine if k < 30 else nxt - 30); nxt += 1
    r2 = nxt; nxt += 1                        # a second and a third tree
    child.append(r2); parent.append(r2)
    for k in range(200):
        child.append(nxt); parent.append(r2 if k < 5 else nxt - 5); nxt += 1
    r3 = nxt + 10
    child.append(r3); parent.append(r3)
    child, parent = np.array(child, np.uint32), np.array(parent, np.uint32)
    mo, mn = np.array([r3 + 5, r3 + 6], np.uint32), np.array([int(child[777]), 999_999], np.uint32)
    tax = O.Taxonomy(child, parent, mo, mn)
    pool = np.concatenate([child, child, child, [0, 0, r3 + 5, r3 + 6, r3 + 7, 2**31]]).astype(np.uint32)
    rng = np.random.default_rng(11)
    for defer, fix in (("1", "1"), ("1", "0"), ("0", "1")):  # (fix: the root-path pairs leave the tile through the fix-up list)
        monkeypatch.setenv("UKM_SETOP_DEFER", defer)
        monkeypatch.setenv("UKM_SETOP_FIX", fix)
        c = L.Context(0)
        c.taxonomy_load(child, parent, mo, mn)
        got = c.lca(pool[:4000], pool[::-1][:4000])
        assert np.array_equal(got, np.array([tax.lca(x, y) for x, y in zip(pool[:4000], pool[::-1][:4000])], np.uint32))
        for n in (50, 6143, 6144, 6145, 120_000):
            A, B = synth_sets(n, 20)
            ta, tb = rng.choice(pool, len(A)).astype(np.uint32), rng.choice(pool, len(B)).astype(np.uint32)
            for op, ofn, fl, kw in ((L.OP_UNION, O.union, 0, {}), (L.OP_INTER, O.inter, 0, {}),
                                    (L.OP_INTER, O.inter, L.F_MIX_TAXID, {"mix_taxid": True}), (L.OP_DIFF, O.diff, 0, {}),
                                    (L.OP_DIFF, O.diff, L.F_CMP_TAXID, {"compare_taxid": True})):
                gk, gt = c.setop2(op, A, B, ta, tb, flags=fl)
                ek, et = ofn([A, B], [ta, tb], tax, **kw)
                assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, defer, n, op, fl)
        # identical sets: every record a match (the queue takes every place the B records leave), all pairs need an LCA
        A = np.arange(1, 20001, dtype=np.uint64) * 5
        ta, tb = rng.choice(child, len(A)).astype(np.uint32), rng.choice(child, len(A)).astype(np.uint32)
        for op, ofn in ((L.OP_UNION, O.union), (L.OP_INTER, O.inter)):
            gk, gt = c.setop2(op, A, A, ta, tb)
            ek, et = ofn([A, A], [ta, tb], tax)
            assert np.array_equal(gk, ek) and np.array_equal(gt, et), (mode, defer, op, "all matched")
        # a result that does not fit: the call fails, nothing is written behind the caller's arrays (the fix-up list holds
        # output positions)
        guard = np.full(len(A) // 2 + 64, 0xABCDEF01, np.uint32)
        with pytest.raises(L.CapacityError):
            c.setop2(L.OP_UNION, A, A, ta, tb, out=np.empty(len(A) // 2, np.uint64), out_taxids=guard[:len(A) // 2])
        assert (guard[len(A) // 2:] == 0xABCDEF01).all()
        c.close()


def test_lca_matches_oracle(ctx, O, tree):
    tax, T = tree
    rng = np.random.default_rng(3)
    a = rng.integers(0, T + 50, 20000).astype(np.uint32)   # includes 0 and unknown ids
    b = rng.integers(0, T + 50, 20000).astype(np.uint32)
    got = ctx.lca(a, b)
    exp = np.array([tax.lca(x, y) for x, y in zip(a, b)], dtype=np.uint32)
    assert np.array_equal(got, exp)


def test_lca_merged_and_forest(ctx, O):
    from unikmer_amd import lib
    c = lib.Context(0)
    child = np.array([1, 2, 3, 4, 5, 10, 11], dtype=np.uint32)
    parent = np.array([1, 1, 1, 2, 2, 10, 10], dtype=np.uint32)
    mo, mn = np.array([7, 8], dtype=np.uint32), np.array([4, 99], dtype=np.uint32)
    c.taxonomy_load(child, parent, mo, mn)
    tax = O.Taxonomy(child, parent, mo, mn)
    ids = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 99, 1000], dtype=np.uint32)
    a, b = np.meshgrid(ids, ids)
    a, b = a.ravel().copy(), b.ravel().copy()
    exp = np.array([tax.lca(x, y) for x, y in zip(a, b)], dtype=np.uint32)
    assert np.array_equal(c.lca(a, b), exp)
    assert c.max_taxid() == 11
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["wide_root", "forest_300", "forest_70000", "bushy", "skewed"])
def test_lca_clade_code_forms(O, L, shape):
    """Round 5: pairs of unrelated taxids are settled by per-id clade codes (ukm_tax.hip: the ancestor at depth <= D as a
    one-byte or two-byte index, D chosen by what fits) and the rest by the root paths.  Every form against the oracle's
    ancestor walk: a root with 400 children (one byte holds the root alone: every pair takes the root paths), a forest of
    300 trees (two-byte codes), one of 70,000 roots (no table at all), a bushy tree (one byte), a skewed one (the one-byte
    cut splits the large kingdom deeper than the small ones); with ids that are zero, unknown, absent inside the range and
    merged."""
    child, parent = [], []
    nxt = [1]

    def new(par=None):
        t = nxt[0]; nxt[0] += 1
        if t % 37 == 0:              # (a gap: an id that is absent inside the range)
            t = nxt[0]; nxt[0] += 1
        child.append(t); parent.append(t if par is None else par)
        return t

    def subtree(par, fan, depth):
        if depth == 0:
            return
        for _ in range(fan):
            subtree(new(par), fan, depth - 1)

    if shape == "wide_root":
        r = new()
        for _ in range(400):
            subtree(new(r), 3, 1)
    elif shape == "forest_300":
        for _ in range(300):
            subtree(new(), 3, 3)
    elif shape == "forest_70000":
        for i in range(70000):
            r = new()
            if i % 100 == 0:
                subtree(r, 2, 2)
    elif shape == "skewed":
        # one large kingdom beside small ones and a chain: the one-byte cut follows the shape (the large kingdom is split
        # three levels down, the small ones stay whole, a node with 300 children is never split)
        r = new()
        big = new(r)
        for _ in range(12):
            subtree(new(big), 7, 3)
        small = new(r)
        subtree(small, 3, 2)
        wide = new(r)
        for _ in range(300):
            new(wide)
        x = new(r)
        for _ in range(30):
            x = new(x)
    else:
        subtree(new(), 6, 5)
    child, parent = np.array(child, dtype=np.uint32), np.array(parent, dtype=np.uint32)
    top = int(child.max())
    mo = np.array([top + 5, top + 6, top + 7], dtype=np.uint32)
    mn = np.array([child[len(child) // 2], child[3], top + 900], dtype=np.uint32)   # (the last one: merged into nothing)
    c = L.Context(0)
    c.taxonomy_load(child, parent, mo, mn)
    tax = O.Taxonomy(child, parent, mo, mn)
    rng = np.random.default_rng(17)
    a = rng.integers(0, top + 12, 30000).astype(np.uint32)
    b = rng.integers(0, top + 12, 30000).astype(np.uint32)
    b[::11] = a[::11]
    # relatives: a node against a node a few ids away (same subtree more often than not)
    a[1::5] = child[rng.integers(0, len(child), len(a[1::5]))]
    b[1::5] = np.minimum(a[1::5] + rng.integers(0, 4, len(a[1::5])).astype(np.uint32), top)
    exp = np.array([tax.lca(x, y) for x, y in zip(a, b)], dtype=np.uint32)
    assert np.array_equal(c.lca(a, b), exp)
    c.close()


def test_failed_taxonomy_load_leaves_the_previous_one(O, L):
    """A load either replaces the taxonomy completely or not at all (round-3 advice): a dump whose dense tables do not fit
    the device's free memory (a sparse huge taxid), a cyclic dump and taxid 0 are refused with an error, and the context
    still answers from the taxonomy it had -- LCA and a union with taxids."""
    c = L.Context(0)
    child = np.array([1, 2, 3, 4, 5], dtype=np.uint32)
    parent = np.array([1, 1, 1, 2, 2], dtype=np.uint32)
    c.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    check = lambda: (c.lca(np.array([4, 4, 3], np.uint32), np.array([5, 3, 0], np.uint32)).tolist() == [2, 1, 0]
                     and c.max_taxid() == 5)
    assert check()
    # largest taxid 2^32 - 2: at least 25 B x 4.3e9 ids = 107 GB of dense tables; make sure that is more than what is free
    import torch
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free - (60 << 30)), dtype=torch.uint8, device="cuda") if free > (60 << 30) else None
    with pytest.raises(L.UkmError) as e:
        c.taxonomy_load(np.array([1, 4294967294], np.uint32), np.array([1, 1], np.uint32))
    assert "renumber" in str(e.value)
    del hog
    torch.cuda.empty_cache()
    assert check()
    with pytest.raises(L.UkmError):
        c.taxonomy_load(np.array([7, 8], np.uint32), np.array([8, 7], np.uint32))      # a cycle
    with pytest.raises(L.UkmError):
        c.taxonomy_load(np.array([0, 2], np.uint32), np.array([1, 1], np.uint32))      # taxid 0 is reserved
    assert check()
    A = np.array([10, 20, 30], np.uint64)
    gk, gt = c.union([A, A[1:]], [np.array([4, 4, 3], np.uint32), np.array([5, 3], np.uint32)])
    ok, ot = O.union([A, A[1:]], [np.array([4, 4, 3], np.uint32), np.array([5, 3], np.uint32)], tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    c.close()


def test_lca_deep_and_ragged_trees(O, L):
    """The root-path table (round 3: 16 bytes per node per 4 levels, LCA = last equal entry of two root paths) against
    the oracle's ancestor walk on shapes that cross chunk boundaries: a chain of 203 nodes, a caterpillar, a forest of
    three trees, ids with gaps, merged ids."""
    c = L.Context(0)
    child, parent = [], []
    # tree 1: chain 100 -> 101 -> ... -> 302 (depth 202), with a leaf hanging off every 7th node (ids 1000 + i)
    child.append(100); parent.append(100)
    for i in range(101, 303):
        child.append(i); parent.append(i - 1)
    for i in range(100, 303, 7):
        child.append(1000 + i); parent.append(i)
    # tree 2: complete ternary tree of depth 5 rooted at 5000
    base, nodes = 5000, sum(3 ** d for d in range(6))
    for t in range(1, nodes + 1):
        child.append(base + t - 1); parent.append(base + (0 if t == 1 else (t - 2) // 3 + 1) - (0 if t == 1 else 1))
    # tree 3: a root that appears only as a parent
    child += [9001, 9002]; parent += [9000, 9001]
    child, parent = np.array(child, dtype=np.uint32), np.array(parent, dtype=np.uint32)
    mo, mn = np.array([50, 51, 52], dtype=np.uint32), np.array([302, 5003, 77777], dtype=np.uint32)
    c.taxonomy_load(child, parent, mo, mn)
    tax = O.Taxonomy(child, parent, mo, mn)
    rng = np.random.default_rng(5)
    pool = np.concatenate([child, [0, 50, 51, 52, 9000, 400, 99999]]).astype(np.uint32)
    a, b = rng.choice(pool, 30000), rng.choice(pool, 30000)
    # many pairs inside the chain (deep common prefixes: several chunks of the table)
    a[:8000] = rng.integers(100, 303, 8000)
    b[:8000] = np.where(rng.random(8000) < 0.5, rng.integers(100, 303, 8000), 1000 + 100 + 7 * rng.integers(0, 29, 8000))
    exp = np.array([tax.lca(x, y) for x, y in zip(a, b)], dtype=np.uint32)
    assert np.array_equal(c.lca(a.astype(np.uint32), b.astype(np.uint32)), exp)
    c.close()


# ---------------------------------------------------------------------------------- n-way
def _files(nfiles, n_universe, p, seed=11):
    j = np.arange(n_universe, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64(np.uint64(SEED) ^ j) & np.uint64((1 << 20) - 1))
    U = np.cumsum(gaps, dtype=np.uint64)
    out = []
    for f in range(nfiles):
        h = splitmix64(np.uint64(seed + 1000 * f) ^ j)
        out.append(U[(h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p])
    return out


@pytest.mark.parametrize("nfiles", [1, 2, 3, 7, 16])
def test_nway_ops_match_oracle(ctx, O, L, tree, nfiles):
    tax, T = tree
    files = _files(nfiles, 60000, 0.7)
    taxs = [taxids_for(f, T, SEED + 13 * i) for i, f in enumerate(files)]
    assert np.array_equal(ctx.union(files), O.union(files))
    assert np.array_equal(ctx.inter(files), O.inter(files))
    assert np.array_equal(ctx.diff(files), O.diff(files))
    for thr in {1, max(1, nfiles // 2), nfiles}:
        assert np.array_equal(ctx.common(files, thr), O.common(files, thr))
    for fn, ofn in [(ctx.union, O.union), (ctx.inter, O.inter), (ctx.diff, O.diff)]:
        gk, gt = fn(files, taxs)
        ok, ot = ofn(files, taxs, tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.common(files, max(1, nfiles - 1), taxs)
    ok, ot = O.common(files, max(1, nfiles - 1), taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    gk, gt = ctx.diff(files, taxs, compare_taxid=True)
    ok, ot = O.diff(files, taxs, tax, compare_taxid=True)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_nway_edge_cases(ctx, O, L):
    files = _files(4, 5000, 0.5)
    e = np.empty(0, np.uint64)
    # union tolerates unsorted inputs with duplicates (hash-map semantics, union.go:186-208)
    rng = np.random.default_rng(5)
    shuffled = [rng.permutation(np.concatenate([f, f[:100]])) for f in files]
    assert np.array_equal(ctx.union(shuffled), O.union(shuffled))
    # the order check is lazy (done by the merges themselves): mixtures of clean, multiset and unsorted
    # streams, an odd count (one stream is carried up the tree unchecked), taxids along
    five = _files(5, 4000, 0.5, seed=23)
    mix5 = [five[0], rng.permutation(five[1]), np.sort(np.concatenate([five[2], five[2][:50]])), five[3], rng.permutation(five[4])]
    assert np.array_equal(ctx.union(mix5), O.union(mix5))
    assert np.array_equal(ctx.union(mix5[:3]), O.union(mix5[:3]))
    assert np.array_equal(ctx.merge_k([five[0], five[1], five[2]], mode=L.UNIQUE), O.union(five[:3]))
    # k-way merge / common with an unsorted stream: the optimistic merge tree reports it and the call falls
    # back to concatenate + sort
    um = [five[0], rng.permutation(five[1]), five[2]]
    assert np.array_equal(ctx.merge_k(um, mode=L.PLAIN), np.sort(np.concatenate(um)))
    assert np.array_equal(ctx.merge_k(um, mode=L.REPEATED), O.merge_k([five[0], np.sort(five[1]), five[2]], mode=O.REPEATED))
    assert np.array_equal(ctx.common(um, 2), O.common(um, 2))
    # inter: an empty LATER file stops the fold and keeps the running result (inter.go:211-217)
    assert np.array_equal(ctx.inter([files[0], files[1], e, files[2]]), O.inter([files[0], files[1], e, files[2]]))
    assert np.array_equal(ctx.inter([files[0], files[1], e, files[2]]), np.intersect1d(files[0], files[1]))
    assert len(ctx.inter([e, files[0]])) == 0
    # diff: unsorted later files, empty files
    sf = [1, 0, 1, 0]
    mixed = [files[0], rng.permutation(files[1]), files[2], rng.permutation(files[3])]
    assert np.array_equal(ctx.diff(mixed, sorted_flags=sf), O.diff(mixed, sorted_flags=sf))
    assert np.array_equal(ctx.diff([files[0], e, files[1]]), O.diff([files[0], e, files[1]]))
    with pytest.raises(L.UnsortedError):
        ctx.diff([rng.permutation(files[0]), files[1]])
    # diff: a first file with duplicate codes keeps every record BETWEEN the files (diff.go:437 mc1 = mc2: a second copy
    # outlives a file that holds the code once and meets the next file) and collapses to one record per code at the end
    dup0 = [np.sort(np.concatenate([files[0], files[0][:700], files[0][:200]])), files[1], files[2], files[3]]
    assert np.array_equal(ctx.diff(dup0), O.diff(dup0))
    assert np.array_equal(ctx.diff(dup0[:3]), O.diff(dup0[:3]))
    assert np.array_equal(ctx.diff(dup0[:2]), O.diff(dup0[:2]))
    # common threshold helper
    assert ctx.common_threshold(4, 0.6) == O.common_threshold(4, 0.6)


# ---------------------------------------------------------------------------------- sort / unique / merge
@pytest.mark.parametrize("n,bits", [(0, 64), (1, 64), (2, 64), (4096, 64), (4097, 62), (100_003, 42), (1_000_000, 62), (300_000, 64)])
def test_sort_u64(ctx, O, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if bits < 64:
        keys &= np.uint64((1 << bits) - 1)
    exp = np.sort(keys)
    got = ctx.sort_u64(keys.copy(), bits)
    assert np.array_equal(got, exp)
    assert np.array_equal(O.sort_u64(keys), exp)


def test_sort_skewed_and_sorted_inputs(ctx):
    n = 500_000
    for keys in (np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64)[::-1].copy(),
                 np.full(n, 12345, dtype=np.uint64), (np.arange(n, dtype=np.uint64) % np.uint64(3)) << np.uint64(40)):
        assert np.array_equal(ctx.sort_u64(keys.copy()), np.sort(keys))


def test_sort_pairs_is_stable(ctx, O):
    rng = np.random.default_rng(1)
    n = 400_000
    keys = rng.integers(0, 5000, n).astype(np.uint64) << np.uint64(20)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = ctx.sort_pairs(keys.copy(), vals.copy(), 62)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(gk, keys[order]) and np.array_equal(gv, vals[order])
    ok, ov = O.sort_pairs(keys, vals)
    assert np.array_equal(gk, ok) and np.array_equal(gv, ov)


@pytest.mark.parametrize("n", [0, 1, 2, 2047, 2048, 2049, 250_000])
def test_unique_modes(ctx, O, L, tree, n):
    tax, T = tree
    rng = np.random.default_rng(n)
    keys = np.sort(rng.integers(0, max(1, n // 3) + 1, n).astype(np.uint64))
    tx = taxids_for(np.arange(n, dtype=np.uint64), T)
    for mode in (L.PLAIN, L.UNIQUE, L.REPEATED, L.REPEATED_CHUNK, L.SINGLETON):
        assert np.array_equal(ctx.unique(keys, mode=mode), O.unique(keys, mode=mode))
        gk, gt = ctx.unique(keys, tx, mode=mode)
        ok, ot = O.unique(keys, tx, mode=mode, tax=tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot)


def test_count_in_one_call(ctx, O, L, genomes):
    """ukm_count = every window -> sort -> the distinct / repeated / singleton set in ONE call (the body of `count`'s Run closure,
    count.go:285-436, and its sort count.go:581) against the oracle's rolling encoder / hasher, its sort and its scans: codes
    for several k, canonical or not, circular records, ntHash with and without the Scaled filter, ragged / short / empty
    records, the README's genome counts (README.md:200-204), a device-resident input, too small an output."""
    rng = np.random.default_rng(66)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 300_000)].copy()
    seq[1000:1040] = ord("N")                                      # degenerate bases collapse (kmers: N -> A)
    off = np.array([0, 10, 10, 5000, 5020, 120_000, 299_990, 300_000], dtype=np.uint64)   # short, empty, long records
    # (a window repeats when the same stretch occurs twice: plant repeats so that -d / -u are not trivial)
    seq[200_000:201_000] = seq[50_000:51_000]
    for k, canonical, circular in ((31, True, False), (21, False, False), (5, True, False), (23, True, True), (32, True, False)):
        w = O.count_windows(seq, off, k, canonical=canonical, circular=circular)
        srt = O.sort_u64(w)
        for mode in (L.UNIQUE, L.REPEATED, L.SINGLETON):
            got = ctx.count(seq, off, k, canonical=canonical, circular=circular, mode=mode)
            assert np.array_equal(got, O.unique(srt, mode=mode)), (k, canonical, circular, mode)
    for k, scale in ((51, 0), (23, 0), (51, 50), (31, 1000)):
        mh = O.max_hash(scale) if scale else 0
        w = O.count_windows(seq, off, k, hashed=True, canonical=True, max_hash=mh)
        srt = O.sort_u64(w)
        for mode in (L.UNIQUE, L.REPEATED, L.SINGLETON):
            assert np.array_equal(ctx.count(seq, off, k, hashed=True, max_hash=mh, mode=mode), O.unique(srt, mode=mode)), (k, scale, mode)
    # the reference's own numbers: distinct canonical 23-mers of the fixture genomes (README.md:200-204)
    from conftest import AMUC, MG1655
    for name, n in ((MG1655, 4546632), (AMUC, 2630905)):
        s, o = genomes(name)
        assert len(ctx.count(s, o, 23)) == n
    # device-resident input and output
    import torch
    ds, do = torch.from_numpy(seq).cuda(), torch.from_numpy(off.view(np.int64)).cuda()
    got = ctx.count(ds, do, 31)
    assert got.is_cuda and np.array_equal(got.cpu().numpy().view(np.uint64), O.unique(O.sort_u64(O.count_windows(seq, off, 31))))
    with pytest.raises(L.CapacityError):
        ctx.count(seq, off, 31, out=np.empty(10, np.uint64))
    with pytest.raises(L.UkmError):
        ctx.count(seq, off, 31, mode=L.PLAIN)
    assert len(ctx.count(seq[:0], np.array([0], np.uint64), 31)) == 0


def test_unique_unsorted_is_an_error(ctx, L):
    with pytest.raises(L.UnsortedError):
        ctx.unique(np.array([3, 1, 2], dtype=np.uint64))


@pytest.mark.parametrize("nstreams", [1, 2, 5])
def test_merge_k(ctx, O, L, tree, nstreams):
    tax, T = tree
    rng = np.random.default_rng(nstreams)
    streams = [np.sort(rng.integers(0, 20000, 30000).astype(np.uint64)) for _ in range(nstreams)]
    taxs = [taxids_for(s + np.uint64(i), T) for i, s in enumerate(streams)]
    for mode in (L.PLAIN, L.UNIQUE, L.REPEATED):
        for final in (True, False):
            assert np.array_equal(ctx.merge_k(streams, mode=mode, final_round=final),
                                  O.merge_k(streams, mode=mode, final_round=final))
            gk, gt = ctx.merge_k(streams, taxs, mode=mode, final_round=final)
            ok, ot = O.merge_k(streams, taxs, mode=mode, final_round=final, tax=tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # C-9: union -s == sort -u == merge -u of sorted chunks
    assert np.array_equal(ctx.merge_k(streams, mode=L.UNIQUE), ctx.union(streams))


# ---------------------------------------------------------------------------------- encode / hash
def _synth_fasta(n_bases, seed=SEED):
    i = np.arange(n_bases, dtype=np.uint64)
    w = splitmix64(np.uint64(seed) ^ (i >> np.uint64(5)))
    code = (w >> (np.uint64(2) * (i & np.uint64(31)))) & np.uint64(3)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code.astype(np.int64)]


@pytest.mark.parametrize("k", [1, 4, 21, 31, 32])
@pytest.mark.parametrize("canonical", [True, False])
def test_encode_kmers(ctx, O, k, canonical):
    bases = _synth_fasta(300_000)
    # ragged records incl. empty ones and ones shorter than k
    cuts = np.array([0, 0, 3, 40, 41, 4096, 4096 + 31, 100_000, 100_010, 299_999, 300_000], dtype=np.uint64)
    got = ctx.encode_kmers(bases, cuts, k, canonical=canonical)
    exp = O.count_windows(bases, cuts, k, hashed=False, canonical=canonical)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("k", [1, 23, 51, 64])
def test_nthash(ctx, O, k):
    bases = _synth_fasta(200_000, SEED + 9)
    cuts = np.array([0, 150, 300, 301, 5000, 5000, 77_777, 200_000], dtype=np.uint64)
    for canonical in (True, False):
        got = ctx.nthash(bases, cuts, k, canonical=canonical)
        exp = O.count_windows(bases, cuts, k, hashed=True, canonical=canonical)
        assert np.array_equal(got, exp)
    mh = O.max_hash(50)
    got = ctx.nthash(bases, cuts, k, max_hash=mh)
    exp = O.count_windows(bases, cuts, k, hashed=True, max_hash=mh)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("force", ["1", None])
def test_nthash_scaled_strip_kernel(ctx, O, monkeypatch, force):
    """The rolling strip kernel behind `count -H --scale N` (count.go:361,373-375): every k phase (k mod 4),
    ragged records incl. empty / shorter-than-k ones and record ends inside a strip's warm-up, non-ACGT bytes
    (zero seed), several tiles, canonical and forward hashes.  force=1 runs it at every scale (a small scale
    overflows its candidate list and must fall back); None = the library's own choice."""
    if force is None:
        monkeypatch.delenv("UKM_NTHASH_STRIP", raising=False)
    else:
        monkeypatch.setenv("UKM_NTHASH_STRIP", force)
    bases = _synth_fasta(1_200_000, SEED + 17).copy()
    bases[1000:1100] = ord("N")
    bases[70_000] = ord("n")
    bases[500_001:500_004] = np.frombuffer(b"acg", dtype=np.uint8)
    bases[900_000] = ord("R")
    cuts = np.array([0, 0, 150, 300, 301, 333, 5000, 5000, 65_535, 65_536, 65_600, 262_144, 262_145, 700_000, 1_199_950,
                     1_200_000], dtype=np.uint64)
    reads = np.arange(0, 1_200_001, 150, dtype=np.uint64)
    for k in (1, 2, 23, 31, 50, 51, 64):
        for scale in (3, 300, 1000, 20000):
            mh = O.max_hash(scale)
            for canonical in (True, False):
                if scale == 3 and (k not in (23, 51) or not canonical):
                    continue
                got = ctx.nthash(bases, cuts, k, canonical=canonical, max_hash=mh)
                exp = O.count_windows(bases, cuts, k, hashed=True, canonical=canonical, max_hash=mh)
                assert np.array_equal(got, exp), (k, scale, canonical)
        mh = O.max_hash(1000)
        assert np.array_equal(ctx.nthash(bases, reads, k, max_hash=mh), O.count_windows(bases, reads, k, hashed=True, max_hash=mh))
    # tiny inputs, a record exactly k long, an input shorter than one strip
    for n in (1, 50, 51, 52, 64, 255, 256, 257, 16_384):
        b = _synth_fasta(n, SEED + n)
        off = np.array([0, n], dtype=np.uint64)
        mh = O.max_hash(4)
        assert np.array_equal(ctx.nthash(b, off, 51, max_hash=mh), O.count_windows(b, off, 51, hashed=True, max_hash=mh)), n
    # low-complexity sequence: every window has the same hash -> all or nothing, far beyond the candidate list
    poly = np.full(300_000, ord("A"), dtype=np.uint8)
    off = np.array([0, 300_000], dtype=np.uint64)
    h0 = O.count_windows(poly[:31], np.array([0, 31], dtype=np.uint64), 31, hashed=True)[0]
    for mh in (int(h0), int(h0) - 1):
        assert np.array_equal(ctx.nthash(poly, off, 31, max_hash=mh), O.count_windows(poly, off, 31, hashed=True, max_hash=mh))
    # a misaligned base pointer (view shifted by one byte)
    buf = np.concatenate([np.zeros(1, np.uint8), bases[:300_000]])
    sh = buf[1:]
    off = np.array([0, 300_000], dtype=np.uint64)
    mh = O.max_hash(500)
    assert np.array_equal(ctx.nthash(sh, off, 31, max_hash=mh), O.count_windows(sh, off, 31, hashed=True, max_hash=mh))


@pytest.mark.parametrize("force", ["1", None])
def test_windows_strip_kernel_long_records(ctx, O, L, monkeypatch, force):
    """The rolling strip kernel for every window of long records (count.go's per-record k-mer loop over a
    chromosome): codes and ntHash, every k phase, canonical / forward, ragged records incl. empty and
    shorter-than-k ones, record ends inside a row block and inside a strip's warm-up, degenerate and lower-case
    bases, several tiles and strip lengths.  force=1 runs it whatever the size; None = the library's own choice (9e6
    bases: the strip kernel for k <= 32, the general kernel for larger k, which switches at 1.7e7 bases)."""
    if force is None:
        monkeypatch.delenv("UKM_WIN_STRIP", raising=False)
    else:
        monkeypatch.setenv("UKM_WIN_STRIP", force)
    n = 9_000_000
    bases = _synth_fasta(n, SEED + 23).copy()
    bases[1000:1100] = ord("N")
    bases[70_000] = ord("n")
    bases[500_001:500_004] = np.frombuffer(b"acg", dtype=np.uint8)
    bases[900_000] = ord("R")
    bases[2_000_000:2_000_040] = np.frombuffer(b"MVHRDWSBYKmvhrdwsbykUuACGTacgtNnACGTACGT", dtype=np.uint8)
    long_cuts = np.array([0, 1_000_003, 1_000_003, 1_000_020, 2_500_001, 4_499_990, n], dtype=np.uint64)
    ragged = np.array([0, 0, 3, 40, 41, 333, 5000, 5000, 65_535, 65_536, 65_543, 65_600, 262_144, 262_145, 700_000,
                       1_199_950, 3_000_000, n - 7, n], dtype=np.uint64)
    cases = [long_cuts, ragged]
    for Ls in (None, "64", "128"):
        if Ls is None:
            monkeypatch.delenv("UKM_WIN_STRIP_L", raising=False)
        else:
            monkeypatch.setenv("UKM_WIN_STRIP_L", Ls)
        for cuts in cases:
            for k in ((1, 2, 7, 8, 9, 21, 31, 32) if Ls is None else (31,)):
                for canonical in (True, False):
                    got = ctx.encode_kmers(bases, cuts, k, canonical=canonical)
                    exp = O.count_windows(bases, cuts, k, hashed=False, canonical=canonical)
                    assert np.array_equal(got, exp), ("codes", Ls, k, canonical, len(cuts))
            for k in ((1, 2, 23, 32, 33, 50, 51, 64) if Ls is None else (51,)):
                for canonical in (True, False):
                    got = ctx.nthash(bases, cuts, k, canonical=canonical)
                    exp = O.count_windows(bases, cuts, k, hashed=True, canonical=canonical)
                    assert np.array_equal(got, exp), ("nthash", Ls, k, canonical, len(cuts))
    monkeypatch.delenv("UKM_WIN_STRIP_L", raising=False)
    # an illegal base is an error only when an emitted window holds it (kmers.ErrIllegalBase)
    bad = bases.copy()
    bad[3_333_333] = ord("*")
    with pytest.raises(L.IllegalBaseError):
        ctx.encode_kmers(bad, long_cuts, 31)
    assert np.array_equal(ctx.nthash(bad, long_cuts, 31), O.count_windows(bad, long_cuts, 31, hashed=True))
    # ... inside a record shorter than k, or in a block next to a record boundary whose windows are not emitted
    bad = bases.copy()
    bad[1_000_010] = ord("*")
    assert np.array_equal(ctx.encode_kmers(bad, long_cuts, 31), O.count_windows(bad, long_cuts, 31))
    bad[1_000_019] = ord("*")
    assert np.array_equal(ctx.encode_kmers(bad, long_cuts, 31), O.count_windows(bad, long_cuts, 31))
    bad[1_000_020] = ord("*")
    with pytest.raises(L.IllegalBaseError):
        ctx.encode_kmers(bad, long_cuts, 31)
    # the minimizer sketch reads the strip kernel's hashes (count.go:316,357)
    for k, w in ((23, 5), (31, 15)):
        got, gpos = ctx.minimizer(bases, long_cuts, k, w, with_pos=True)
        exp, epos = _oracle_minimizer_records(O, bases, long_cuts, k, w)
        assert np.array_equal(got, exp) and np.array_equal(gpos, epos)
    # more records under one tile than the kernel's table holds: falls back to the general kernel
    many = np.concatenate([np.arange(0, 40_000, 100, dtype=np.uint64), np.array([n], dtype=np.uint64)])
    assert np.array_equal(ctx.encode_kmers(bases, many, 21), O.count_windows(bases, many, 21))
    # a misaligned base pointer
    buf = np.concatenate([np.zeros(1, np.uint8), bases])
    off = np.array([0, n], dtype=np.uint64)
    assert np.array_equal(ctx.encode_kmers(buf[1:], off, 31), O.count_windows(buf[1:], off, 31))


def test_circular_and_reads(ctx, O):
    bases = _synth_fasta(50_000, SEED + 1)
    one = np.array([0, 50_000], dtype=np.uint64)
    for k in (5, 31):
        assert np.array_equal(ctx.encode_kmers(bases, one, k, circular=True),
                              O.count_windows(bases, one, k, circular=True))
        assert np.array_equal(ctx.nthash(bases, one, k, circular=True),
                              O.count_windows(bases, one, k, hashed=True, circular=True))
    reads = np.arange(0, 50_001, 150, dtype=np.uint64)   # 150 bp reads
    assert np.array_equal(ctx.encode_kmers(bases, reads, 31), O.count_windows(bases, reads, 31))
    assert np.array_equal(ctx.nthash(bases, reads, 51), O.count_windows(bases, reads, 51, hashed=True))
    small = np.arange(0, 1001, 7, dtype=np.uint64)       # many records per tile, some < k
    assert np.array_equal(ctx.encode_kmers(bases[:1001], small, 5, circular=True),
                          O.count_windows(bases[:1001], small, 5, circular=True))


def test_illegal_and_degenerate_bases(ctx, O, L):
    seq = np.frombuffer(b"ACGTNNACGTRYKMacgtuACGTACGTACGT", dtype=np.uint8)
    off = np.array([0, len(seq)], dtype=np.uint64)
    assert np.array_equal(ctx.encode_kmers(seq, off, 5), O.count_windows(seq, off, 5))
    assert np.array_equal(ctx.nthash(seq, off, 5), O.count_windows(seq, off, 5, hashed=True))
    bad = np.frombuffer(b"ACGTACGT*ACGTACGT", dtype=np.uint8)
    with pytest.raises(L.IllegalBaseError):
        ctx.encode_kmers(bad, np.array([0, len(bad)], dtype=np.uint64), 4)


def _oracle_minimizer_records(O, bases, cuts, k, w, circular=False, max_hash=0):
    hs, ps = [], []
    for r in range(len(cuts) - 1):
        seq = bases[int(cuts[r]):int(cuts[r + 1])]
        try:
            h, p = O.minimizer(seq, k, w, circular=circular)
        except ValueError:     # ErrShortSeq: record skipped (count.go:323-328)
            continue
        if max_hash:
            keep = h <= np.uint64(max_hash)
            h, p = h[keep], p[keep]
        hs.append(h); ps.append(p)
    if not hs:
        return np.empty(0, np.uint64), np.empty(0, np.uint64)
    return np.concatenate(hs), np.concatenate(ps)


@pytest.mark.parametrize("k,w", [(23, 5), (31, 15), (5, 1), (7, 2), (51, 40), (11, 300), (21, 1024)])
def test_minimizer(ctx, O, k, w):
    bases = _synth_fasta(120_000, SEED + 4)
    cuts = np.array([0, 150, 300, 301, 1024, 2047, 2048 + w, 5000, 5000, 77_777, 120_000], dtype=np.uint64)
    for circular in (False, True):
        got, gpos = ctx.minimizer(bases, cuts, k, w, circular=circular, with_pos=True)
        exp, epos = _oracle_minimizer_records(O, bases, cuts, k, w, circular=circular)
        assert np.array_equal(got, exp)
        assert np.array_equal(gpos, epos)
    mh = O.max_hash(7)
    got = ctx.minimizer(bases, cuts, k, w, max_hash=mh)
    exp, _ = _oracle_minimizer_records(O, bases, cuts, k, w, max_hash=mh)
    assert np.array_equal(got, exp)


def test_minimizer_low_complexity_ties(ctx, O):
    """Runs of equal hashes (homopolymers, short repeats) exercise the leftmost-minimum rule."""
    seq = np.frombuffer((b"A" * 300 + b"ACACACACAC" * 40 + b"T" * 200 + b"ACGTTGCA" * 50), dtype=np.uint8)
    off = np.array([0, len(seq)], dtype=np.uint64)
    for k, w in ((5, 3), (9, 8), (15, 20)):
        got, gpos = ctx.minimizer(seq, off, k, w, with_pos=True)
        exp, epos = O.minimizer(seq, k, w)
        assert np.array_equal(got, exp) and np.array_equal(gpos, epos)


def test_minimizer_rejects_large_w(ctx, L):
    seq = np.frombuffer(b"ACGT" * 1000, dtype=np.uint8)
    with pytest.raises(L.UkmError):
        ctx.minimizer(seq, np.array([0, len(seq)], dtype=np.uint64), 21, 5000)


# ---------------------------------------------------------------------------------- KATs through the GPU
def test_kat_count_sort_unique_genomes(ctx, L, genomes):
    """README.md:200-204,270-278 through the HIP path: count -k 23 -K -s, then union/inter/diff."""
    sets = {}
    for name, expected in ((MG1655, 4546632), (IAI39, 4902266), (AMUC, 2630905)):
        bases, off = genomes(name)
        codes = ctx.encode_kmers(bases, off, 23, canonical=True)
        ctx.sort_u64(codes, 46)
        sets[name] = ctx.unique(codes, mode=L.UNIQUE)
        assert len(sets[name]) == expected
    assert [int(c) for c in sets[MG1655][:3]] == [87360378, 94155581, 98566170]
    a, b = sets[IAI39], sets[MG1655]
    assert len(ctx.union([a, b])) == 6872728
    assert len(ctx.inter([a, b])) == 2576170
    assert len(ctx.diff([a, b])) == 2326096
    assert len(ctx.merge_k([a, b], mode=L.REPEATED)) == 2576170
    assert len(ctx.common([a, b], 2)) == 2576170


def test_kat_scaled_minhash(ctx, L, genomes):
    """analysis/distance/README.md:9: count -k 31 -K -s -H -D 15 on MG1655 -> 586,734."""
    bases, off = genomes(MG1655)
    h = ctx.nthash(bases, off, 31, canonical=True, max_hash=ctx.max_hash(15))
    ctx.sort_u64(h)
    assert len(ctx.unique(h)) == 586734


def test_kat_minimizer(ctx, L, genomes):
    """README.md:174,189-199 (count -k 23 -W 5 -H -K -l -> 860,900 minimizers, first positions 2,5,6,9,13)
    and analysis/distance/README.md:8 (MG1655 -k 31 -W 15 -> 549,963 distinct)."""
    bases, off = genomes(AMUC)
    h, pos = ctx.minimizer(bases, off, 23, 5, with_pos=True)
    assert len(h) == 860900
    assert [int(p) for p in pos[:5]] == [2, 5, 6, 9, 13]
    assert [int(x) for x in np.sort(h)[:3]] == [1210726578792, 2286899379883, 3542156397282]
    bases, off = genomes(MG1655)
    h = ctx.minimizer(bases, off, 31, 15)
    ctx.sort_u64(h)
    assert len(ctx.unique(h)) == 549963


# ---------------------------------------------------------------------------------- device-resident tensors
def test_device_tensors_roundtrip(ctx, O, L):
    import torch
    A, B = synth_sets(400_000, 22)
    dA = torch.from_numpy(A.view(np.int64)).cuda()
    dB = torch.from_numpy(B.view(np.int64)).cuda()
    out = torch.empty(len(A) + len(B), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    u = ctx.setop2(L.OP_UNION, dA, dB, out=out)
    assert u.is_cuda and np.array_equal(u.cpu().numpy().view(np.uint64), O.union([A, B]))
    i = ctx.setop2(L.OP_INTER, dA, dB)
    assert np.array_equal(i.cpu().numpy().view(np.uint64), O.inter([A, B]))
    # sort on device, in place
    rng = np.random.default_rng(0)
    k = rng.integers(0, 2**62, 1_000_000, dtype=np.uint64)
    dk = torch.from_numpy(k.view(np.int64)).cuda()
    torch.cuda.synchronize()
    ctx.sort_u64(dk, 62)
    assert np.array_equal(dk.cpu().numpy().view(np.uint64), np.sort(k))
    assert ctx.last_kernel_ms() > 0


def test_setop2_device_views_every_alignment(ctx, O, L):
    """Interior tiles are staged by LDS-DMA in 16-byte units (round 3), so the kernel lays its tile out by the PARITY of
    every input's 8-byte address; the outputs go out in 16-byte stores where their address allows.  Device views that
    start on every combination of odd / even 8-byte slots (inputs and output), at a size with hundreds of interior
    tiles, must give the oracle's result."""
    import torch
    A, B = synth_sets(4_000_000, 22)
    u, i, d = O.union([A, B]), O.inter([A, B]), O.diff([A, B])
    dev = torch.device("cuda", 0)
    pad = 3
    bufa = torch.zeros(len(A) + pad, dtype=torch.int64, device=dev)
    bufb = torch.zeros(len(B) + pad, dtype=torch.int64, device=dev)
    bufo = torch.zeros(len(A) + len(B) + pad, dtype=torch.int64, device=dev)
    for oa in (0, 1):
        for ob in (0, 1, 2):
            for oo in (0, 1):
                ta, tb = bufa[oa:oa + len(A)], bufb[ob:ob + len(B)]
                ta.copy_(torch.from_numpy(A.view(np.int64)))
                tb.copy_(torch.from_numpy(B.view(np.int64)))
                torch.cuda.synchronize()
                for op, exp in ((L.OP_UNION, u), (L.OP_INTER, i), (L.OP_DIFF, d)):
                    got = ctx.setop2(op, ta, tb, out=bufo[oo:])
                    assert got.data_ptr() == bufo.data_ptr() + 8 * oo
                    assert np.array_equal(got.cpu().numpy().view(np.uint64), exp), (oa, ob, oo, op)


def test_partition_points(ctx):
    A, _ = synth_sets(100_000, 22)
    sp = np.array([0, A[10], A[10] + np.uint64(1), A[-1], 2**63], dtype=np.uint64)
    assert np.array_equal(ctx.partition_points(A, sp), np.searchsorted(A, sp, side="left").astype(np.uint64))


def test_ticketed_fallback_kernel(O, L, monkeypatch):
    """The dispatch-order independent variant (used if the fast kernel's watchdog ever fires)."""
    monkeypatch.setenv("UKM_FORCE_TICKET", "1")
    c = L.Context(0)
    A, B = synth_sets(700_000, 22)
    assert np.array_equal(c.setop2(L.OP_UNION, A, B), O.union([A, B]))
    assert np.array_equal(c.setop2(L.OP_INTER, A, B), O.inter([A, B]))
    assert np.array_equal(c.setop2(L.OP_DIFF, A, B), O.diff([A, B]))
    c.close()


def test_two_contexts_from_two_threads(O, L):
    """The boundary is re-entrant (SURVEY.md §8(b) Threading): one ukm_ctx per calling OS thread, no global
    mutable state.  Two threads hammer set operations / sorts through their own contexts concurrently."""
    import threading
    results, errors = {}, []

    def work(tid):
        try:
            c = L.Context(0)
            rng = np.random.default_rng(100 + tid)
            for it in range(6):
                a = np.unique(rng.integers(0, 1 << 40, 300_000).astype(np.uint64))
                b = np.unique(rng.integers(0, 1 << 40, 200_000).astype(np.uint64))
                u, i = c.setop2(L.OP_UNION, a, b), c.setop2(L.OP_INTER, a, b)
                x = rng.integers(0, 1 << 62, 250_000).astype(np.uint64)
                s = c.sort_u64(x.copy(), 62)
                results[(tid, it)] = (np.array_equal(u, np.union1d(a, b)) and np.array_equal(i, np.intersect1d(a, b))
                                      and np.array_equal(s, np.sort(x)))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(results) == 12 and all(results.values())


def test_full_64bit_keys_unsigned_order(ctx, O, L):
    """Hashed k-mers (ntHash, -H) use all 64 bits: every comparison must be unsigned."""
    rng = np.random.default_rng(9)
    a = np.unique(rng.integers(0, 1 << 64, 200_000, dtype=np.uint64))
    b = np.unique(np.concatenate([rng.integers(0, 1 << 64, 150_000, dtype=np.uint64), a[::3]]))
    assert (a >> np.uint64(63)).any() and (b >> np.uint64(63)).any()
    assert np.array_equal(ctx.setop2(L.OP_UNION, a, b), np.union1d(a, b))
    assert np.array_equal(ctx.setop2(L.OP_INTER, a, b), np.intersect1d(a, b))
    assert np.array_equal(ctx.setop2(L.OP_DIFF, a, b), np.setdiff1d(a, b))
    cat = np.concatenate([b, a])
    assert np.array_equal(ctx.sort_u64(cat.copy(), 64), np.sort(cat))
    assert np.array_equal(ctx.merge_k([a, b], mode=L.PLAIN), np.sort(cat))
    assert np.array_equal(ctx.merge_k([a, b], mode=L.REPEATED), np.intersect1d(a, b))
    assert np.array_equal(ctx.common([a, b, a], 3), np.intersect1d(a, b))
    pts = ctx.partition_points(a, np.array([0, 1 << 63, (1 << 64) - 1], dtype=np.uint64))
    assert [int(x) for x in pts] == [int(np.searchsorted(a, np.uint64(v))) for v in (0, 1 << 63, (1 << 64) - 1)]


def test_sharded_setop_over_rccl_world1(ctx, O, L):
    """unikmer_amd/dist.py end to end on the GPU with the `nccl` (= RCCL) backend and a world of one
    rank: the all-to-all-v of int64 device tensors, the GPU cut points (ukm_partition_points) and the
    per-rank n-way op all run; the 2-rank behaviour of the same code is covered on CPU over gloo."""
    import socket
    import torch
    import torch.distributed as dist
    from unikmer_amd import dist as ud
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(21)
        files = [np.unique(rng.integers(0, 1 << 42, 60_000, dtype=np.uint64)) for _ in range(3)]
        dfiles = [torch.from_numpy(f.view(np.int64)).to(dev) for f in files]
        dctx = L.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        for op, ref in (("union", O.union), ("inter", O.inter), ("diff", O.diff)):
            got = ud.sharded_setop(dctx, op, dfiles, 42)
            got = got[0] if isinstance(got, tuple) else got
            assert np.array_equal(got.cpu().numpy().view(np.uint64), np.sort(ref(files)))
        # the count path: unsorted codes -> distributed sort / distinct set
        x = rng.integers(0, 1 << 42, 200_000, dtype=np.uint64)
        x[:1000] = x[1000:2000]
        dk = torch.from_numpy(x.copy().view(np.int64)).to(dev)
        assert np.array_equal(ud.sharded_sort(dctx, dk.clone(), 42).cpu().numpy().view(np.uint64), np.sort(x))
        assert np.array_equal(ud.sharded_count(dctx, dk.clone(), 42).cpu().numpy().view(np.uint64), np.unique(x))
        tv = torch.arange(len(x), dtype=torch.int32, device=dev)
        sk, st = ud.sharded_sort(dctx, dk.clone(), 42, tv.clone())
        assert np.array_equal(x[st.cpu().numpy()], sk.cpu().numpy().view(np.uint64))
    finally:
        dist.destroy_process_group()


def test_c_example_client_runs(tmp_path):
    """examples/count_union.c (plain C99 over the header) on the GPU: encode -> sort -> unique -> union/inter."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "count_union")
    libdir = os.path.join(root, "unikmer_amd")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "count_union.c"),
                    "-L", libdir, "-lunikmer_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    # expected values from a direct Python evaluation of the same two sequences
    def canon_set(s, k=11):
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        out = set()
        for i in range(len(s) - k + 1):
            f = s[i:i + k]
            r_ = "".join(comp[c] for c in reversed(f))
            enc = lambda x: int("".join(str("ACGT".index(c)) for c in x), 4)
            out.add(min(enc(f), enc(r_)))
        return out
    a = canon_set("ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTAGCTAGGATCCGATCGTTAGC")
    b = canon_set("TTGCAAGGCTTAACCGGTTACGTTTTTTTTGATCGGCTAGCTAGGATCC")
    assert r.stdout.strip() == "k=11 |A|=%d |B|=%d |A u B|=%d |A n B|=%d" % (len(a), len(b), len(a | b), len(a & b))


def test_setop2_partition_on_skewed_inputs(ctx, O, L):
    """The merge-path partition brackets every split around an interpolated position (evenly spread keys); inputs
    that are anything but evenly spread must only lose the two probes: disjoint value ranges in both orders, a dense
    cluster of one set inside the other's range, one value repeated (multisets), sets of very different size, and a
    geometric spread — several hundred tiles each, i.e. both partition levels, every operation against numpy."""
    rng = np.random.default_rng(77)
    n = 3_000_000

    def uniq_sorted(x):
        return np.unique(x.astype(np.uint64))

    lowhalf = uniq_sorted(rng.integers(0, 1 << 40, n))
    highhalf = uniq_sorted(rng.integers(1 << 41, 1 << 42, n))
    wide = uniq_sorted(rng.integers(0, 1 << 62, n))
    cluster = uniq_sorted((1 << 61) + rng.integers(0, 1 << 24, n))
    geometric = uniq_sorted((rng.random(n) ** 8 * float(1 << 62)).astype(np.uint64))
    tiny = uniq_sorted(rng.integers(0, 1 << 62, 1000))
    cases = [(lowhalf, highhalf), (highhalf, lowhalf), (wide, cluster), (cluster, wide), (geometric, wide),
             (wide, geometric), (tiny, wide), (wide, tiny), (geometric, geometric[::3].copy())]
    for a, b in cases:
        assert np.array_equal(ctx.setop2(L.OP_UNION, a, b), np.union1d(a, b))
        assert np.array_equal(ctx.setop2(L.OP_INTER, a, b), np.intersect1d(a, b, assume_unique=True))
        assert np.array_equal(ctx.setop2(L.OP_DIFF, a, b), np.setdiff1d(a, b, assume_unique=True))
    # multisets: long runs of one value straddling many tiles (oracle semantics, util of union/inter/diff on multisets)
    m1 = np.sort(np.concatenate([np.full(1_500_000, 7, np.uint64), wide[:1_000_000]]))
    m2 = np.sort(np.concatenate([np.full(900_000, 7, np.uint64), cluster[:1_200_000], np.full(300_000, 1 << 61, np.uint64)]))
    assert np.array_equal(ctx.setop2(L.OP_UNION, m1, m2), O.union([m1, m2]))
    assert np.array_equal(ctx.setop2(L.OP_INTER, m1, m2), O.inter([m1, m2]))
    assert np.array_equal(ctx.setop2(L.OP_DIFF, m1, m2), O.diff([m1, m2]))
