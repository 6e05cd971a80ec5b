"""Per-FILE taxids (round 5): the .unik header's global taxid -- `unikmer count -t 511145`, README.md:170,
count.go:466-468; the reader hands it out with every record and union.go:187-201 / inter.go:190,211-239 /
diff.go:404-409 / common.go:262-266 / util-sort.go fold it like a per-record taxid -- passed to the C ABI as ONE
number per stream (ukm_*_ft).  Every case is compared, bit-exact, with the CPU oracle fed the EXPANDED arrays: taxids
on a complete tree, on a forest with merged / zero / unknown ids, streams with one taxid per file mixed with streams
that carry one per record and streams with none, every n-way route forced in turn."""
import numpy as np
import pytest

from conftest import splitmix64, synth_tree

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572


def _universe(n, gap_bits=24, seed=SEED):
    j = np.arange(n, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64(np.uint64(seed) ^ j) & np.uint64((1 << gap_bits) - 1))
    return np.cumsum(gaps, dtype=np.uint64)


def _member(n, f, p, seed):
    h = splitmix64(np.uint64(seed + 1000 * (f + 1)) ^ np.arange(n, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p


def _taxids(codes, ids, salt):
    """a per-record taxid drawn from `ids` by a hash of the code"""
    ids = np.asarray(ids, dtype=np.uint32)
    return ids[(splitmix64(np.uint64(SEED + 2 + salt) ^ codes) % np.uint64(len(ids))).astype(np.int64)]


def _expand(files, taxs):
    """what the reference's reader hands out: the file's taxid with every record"""
    out = []
    for f, t in zip(files, taxs):
        if t is None:
            out.append(None)
        elif isinstance(t, (int, np.integer)):
            out.append(np.full(len(f), int(t), np.uint32))
        else:
            out.append(t)
    return out


@pytest.fixture(scope="module", params=["tree", "forest"])
def env(request):
    from oracle import oracle as O
    from unikmer_amd import lib as L
    ctx = L.Context(0)
    if request.param == "tree":
        child, parent = synth_tree(5, 8)
        ctx.taxonomy_load(child, parent)
        tax = O.Taxonomy(child, parent)
        T = len(child)
        ids = np.arange(1, T + 1, dtype=np.uint32)
        # a few deep relatives so that LCAs are not all the root
        pool = np.concatenate([ids[-64:], ids[:9], ids[100:110]])
    else:
        # two trees, merged ids (7 -> 4, 8 -> 99 which does not exist), and the pool also holds 0 and unknown ids
        child = np.array([1, 2, 3, 4, 5, 6, 10, 11, 12, 13], dtype=np.uint32)
        parent = np.array([1, 1, 1, 2, 2, 4, 10, 10, 11, 11], dtype=np.uint32)
        mo, mn = np.array([7, 8], dtype=np.uint32), np.array([4, 99], dtype=np.uint32)
        ctx.taxonomy_load(child, parent, mo, mn)
        tax = O.Taxonomy(child, parent, mo, mn)
        pool = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 99, 1000, 1001], dtype=np.uint32)
    yield O, L, ctx, tax, pool, request.param
    ctx.close()


def _eq(got, exp, what=None):
    gk, gt = got
    ek, et = exp
    assert np.array_equal(gk, ek), what
    assert np.array_equal(gt, et), what


# ---------------------------------------------------------------------------------------------- 2-way
@pytest.mark.parametrize("n", [3, 700, 9_800, 120_000, 1_400_000])
def test_setop2_two_file_taxids(env, n):
    """both streams carry ONE taxid: the plain-key kernel with the taxid epilogue (CT), all four operations and both
    flag rules, over every pair of a small pool of ids (equal, nested, unrelated, merged, zero, unknown)"""
    O, L, ctx, tax, pool, kind = env
    U = _universe(n, 22)
    m = splitmix64(np.uint64(SEED + 1) ^ np.arange(n, dtype=np.uint64)) & np.uint64(3)
    A, B = U[(m == 0) | (m >= 2)], U[(m == 1) | (m >= 2)]
    pairs = [(int(a), int(b)) for a in pool[:6] for b in pool[:6]] if n <= 700 else \
        [(int(pool[i]), int(pool[(5 * i + 3) % len(pool)])) for i in range(0, len(pool), 3)] + [(int(pool[1]), int(pool[1]))]
    for ca, cb in pairs:
        if ca == 0 and cb == 0:
            continue
        ta, tb = np.full(len(A), ca, np.uint32), np.full(len(B), cb, np.uint32)
        _eq(ctx.setop2(L.OP_UNION, A, B, ca, cb), O.union([A, B], [ta, tb], tax), ("union", ca, cb))
        _eq(ctx.setop2(L.OP_INTER, A, B, ca, cb), O.inter([A, B], [ta, tb], tax), ("inter", ca, cb))
        _eq(ctx.setop2(L.OP_INTER, A, B, ca, cb, flags=L.F_MIX_TAXID), O.inter([A, B], [ta, tb], tax, mix_taxid=True), ("mix", ca, cb))
        _eq(ctx.setop2(L.OP_DIFF, A, B, ca, cb), O.diff([A, B], [ta, tb], tax), ("diff", ca, cb))
        _eq(ctx.setop2(L.OP_DIFF, A, B, ca, cb, flags=L.F_CMP_TAXID), O.diff([A, B], [ta, tb], tax, compare_taxid=True), ("diff -t", ca, cb))


def test_setop2_file_taxid_beside_per_record_taxids(env):
    """one stream with a taxid per record, the other with one for the file: the taxid kernel fills the constant in"""
    O, L, ctx, tax, pool, kind = env
    for n in (50, 30_000, 400_000):
        U = _universe(n, 22)
        m = splitmix64(np.uint64(SEED + 9) ^ np.arange(n, dtype=np.uint64)) & np.uint64(3)
        A, B = U[(m == 0) | (m >= 2)], U[(m == 1) | (m >= 2)]
        ta, tb = _taxids(A, pool, 1), _taxids(B, pool, 2)
        for c in (int(pool[2]), int(pool[-1]), int(pool[5])):
            ca, cb = np.full(len(A), c, np.uint32), np.full(len(B), c, np.uint32)
            _eq(ctx.setop2(L.OP_UNION, A, B, ta, c), O.union([A, B], [ta, cb], tax))
            _eq(ctx.setop2(L.OP_UNION, A, B, c, tb), O.union([A, B], [ca, tb], tax))
            _eq(ctx.setop2(L.OP_INTER, A, B, c, tb), O.inter([A, B], [ca, tb], tax))
            _eq(ctx.setop2(L.OP_INTER, A, B, ta, c, flags=L.F_MIX_TAXID), O.inter([A, B], [ta, cb], tax, mix_taxid=True))
            _eq(ctx.setop2(L.OP_DIFF, A, B, c, tb, flags=L.F_CMP_TAXID), O.diff([A, B], [ca, tb], tax, compare_taxid=True))
            _eq(ctx.setop2(L.OP_DIFF, A, B, ta, c, flags=L.F_CMP_TAXID), O.diff([A, B], [ta, cb], tax, compare_taxid=True))
            # a file taxid beside a stream with no taxid information at all
            z = np.zeros(len(B), np.uint32)
            _eq(ctx.setop2(L.OP_UNION, A, B, c, None), O.union([A, B], [ca, z], tax))
            _eq(ctx.setop2(L.OP_INTER, A, B, c, None, flags=L.F_MIX_TAXID), O.inter([A, B], [ca, z], tax, mix_taxid=True))


def test_setop2_file_taxids_multisets_and_devices(env):
    """duplicates inside an input (the rank path) with one taxid per file; device tensors at every 8-byte alignment (the
    epilogue's stores start wherever the look-back put the tile)"""
    import torch
    O, L, ctx, tax, pool, kind = env
    rng = np.random.default_rng(5)
    U = _universe(40_000, 22)
    A = np.sort(np.concatenate([U[::2], U[:3000:7], U[:3000:7]]))
    B = np.sort(np.concatenate([U[::3], U[100:900], U[100:900:2]]))
    ca, cb = int(pool[3]), int(pool[4])
    ta, tb = np.full(len(A), ca, np.uint32), np.full(len(B), cb, np.uint32)
    _eq(ctx.setop2(L.OP_UNION, A, B, ca, cb), O.union([A, B], [ta, tb], tax))
    _eq(ctx.setop2(L.OP_INTER, A, B, ca, cb), O.inter([A, B], [ta, tb], tax))
    _eq(ctx.setop2(L.OP_DIFF, A, B, ca, cb), O.diff([A, B], [ta, tb], tax))
    _eq(ctx.setop2(L.OP_DIFF, A, B, ca, cb, flags=L.F_CMP_TAXID), O.diff([A, B], [ta, tb], tax, compare_taxid=True))
    A, B = U[_member(len(U), 1, 0.6, 3)], U[_member(len(U), 2, 0.6, 3)]
    ta, tb = np.full(len(A), ca, np.uint32), np.full(len(B), cb, np.uint32)
    exp = O.union([A, B], [ta, tb], tax)
    dA = torch.from_numpy(np.concatenate([np.zeros(4, np.uint64), A]).view(np.int64)).cuda()
    dB = torch.from_numpy(np.concatenate([np.zeros(4, np.uint64), B]).view(np.int64)).cuda()
    for sa in range(3):
        for so in range(4):
            out = torch.empty(len(A) + len(B) + 8, dtype=torch.int64, device="cuda")
            tout = torch.empty(len(A) + len(B) + 8, dtype=torch.int32, device="cuda")
            gk, gt = ctx.setop2(L.OP_UNION, dA[4 - sa:][sa:], dB[4:], ca, cb, out=out[so:], out_taxids=tout[so:])
            assert np.array_equal(gk.cpu().numpy().view(np.uint64), exp[0]) and np.array_equal(gt.cpu().numpy().view(np.uint32), exp[1])


# ---------------------------------------------------------------------------------------------- n-way
def _files(nfiles, n_universe, p, seed=11):
    U = _universe(n_universe, 20)
    return [U[_member(len(U), f, p, seed)] for f in range(nfiles)]


def _tax_shapes(files, pool, rng):
    """(name, taxids_list) shapes: every file its own taxid; all the same taxid; files with one taxid beside files with
    one per record and files with none"""
    n = len(files)
    per_file = [int(pool[(3 * i + 1) % len(pool)]) for i in range(n)]
    yield "per file", per_file
    yield "same", [int(pool[2])] * n
    nested = [int(pool[1 + (i % 3)]) for i in range(n)]
    yield "few", nested
    mixed = []
    for i, f in enumerate(files):
        r = i % 4
        mixed.append(per_file[i] if r == 0 else (_taxids(f, pool, i) if r == 1 else (None if r == 2 else 0)))
    yield "mixed", mixed


@pytest.mark.parametrize("nfiles", [1, 2, 3, 7, 16, 45])
def test_nway_file_taxids_match_oracle(env, nfiles):
    O, L, ctx, tax, pool, kind = env
    rng = np.random.default_rng(nfiles)
    files = _files(nfiles, 30_000, 0.7)
    for name, taxs in _tax_shapes(files, pool, rng):
        ex = _expand(files, taxs)
        if all(t is None or (isinstance(t, int) and t == 0) for t in taxs):
            continue
        what = (kind, nfiles, name)
        _eq(ctx.union(files, taxs), O.union(files, ex, tax), ("union",) + what)
        _eq(ctx.inter(files, taxs), O.inter(files, ex, tax), ("inter",) + what)
        _eq(ctx.inter(files, taxs, mix_taxid=True), O.inter(files, ex, tax, mix_taxid=True), ("inter mix",) + what)
        _eq(ctx.diff(files, taxs), O.diff(files, ex, tax), ("diff",) + what)
        _eq(ctx.diff(files, taxs, compare_taxid=True), O.diff(files, ex, tax, compare_taxid=True), ("diff -t",) + what)
        for thr in sorted({1, 2, max(1, nfiles // 2), max(1, nfiles - 1), nfiles}):
            if thr <= nfiles:
                _eq(ctx.common(files, thr, taxs), O.common(files, thr, ex, tax), ("common", thr) + what)
        for mode in (L.PLAIN, L.UNIQUE, L.REPEATED):
            for final in (True, False):
                gk, gt = ctx.merge_k(files, taxs, mode=mode, final_round=final)
                ok, ot = O.merge_k(files, ex, mode=mode, final_round=final, tax=tax)
                assert np.array_equal(gk, ok), ("merge", mode, final) + what
                if mode == L.PLAIN:
                    # equal codes keep stream order in both, but the oracle's heap pops ties in its own order: compare the
                    # taxids of every run as multisets
                    o1 = np.lexsort((gt, gk))
                    o2 = np.lexsort((ot, ok))
                    assert np.array_equal(gt[o1], ot[o2]), ("merge plain",) + what
                else:
                    assert np.array_equal(gt, ot), ("merge", mode, final) + what


def test_nway_file_taxids_quirks(env):
    """inter: an empty LATER file ends the fold, the taxids of the files behind it play no part (inter.go:211-217); diff:
    unsorted later files, empty files, a first file with duplicates; diff -t where every later file is harmless"""
    O, L, ctx, tax, pool, kind = env
    files = _files(6, 20_000, 0.6, seed=5)
    e = np.empty(0, np.uint64)
    cts = [int(pool[(2 * i + 1) % len(pool)]) for i in range(7)]
    fl = [files[0], files[1], files[2], e, files[3], files[4]]
    tx = cts[:6]
    _eq(ctx.inter(fl, tx), O.inter(fl, _expand(fl, tx), tax))
    _eq(ctx.inter(fl, tx, mix_taxid=True), O.inter(fl, _expand(fl, tx), tax, mix_taxid=True))
    fl2 = [files[0], e, files[1], files[2]]
    _eq(ctx.diff(fl2, cts[:4]), O.diff(fl2, _expand(fl2, cts[:4]), tax))
    _eq(ctx.diff(fl2, cts[:4], compare_taxid=True), O.diff(fl2, _expand(fl2, cts[:4]), tax, compare_taxid=True))
    rng = np.random.default_rng(8)
    sf = [1, 0, 1, 0, 1]
    mixed = [files[0], rng.permutation(files[1]), files[2], rng.permutation(files[3]), files[4]]
    _eq(ctx.diff(mixed, cts[:5], sorted_flags=sf), O.diff(mixed, _expand(mixed, cts[:5]), tax, sorted_flags=sf))
    _eq(ctx.diff(mixed, cts[:5], compare_taxid=True, sorted_flags=sf),
        O.diff(mixed, _expand(mixed, cts[:5]), tax, compare_taxid=True, sorted_flags=sf))
    dup0 = [np.sort(np.concatenate([files[0], files[0][:500]])), files[1], files[2], files[3], files[4]]
    _eq(ctx.diff(dup0, cts[:5]), O.diff(dup0, _expand(dup0, cts[:5]), tax))
    _eq(ctx.inter(dup0, cts[:5]), O.inter(dup0, _expand(dup0, cts[:5]), tax))
    # every later file carries the first file's own taxid: diff -t takes nothing away
    same = [cts[1]] * 5
    _eq(ctx.diff(files[:5], same, compare_taxid=True), O.diff(files[:5], _expand(files[:5], same), tax, compare_taxid=True))
    gk, gt = ctx.diff(files[:5], same, compare_taxid=True)
    assert np.array_equal(gk, files[0])
    # common over all files with a duplicate inside a later file: the code can reach the count without being in every file
    dl = [files[0], np.sort(np.concatenate([files[1], files[1][:3000]])), files[2], files[3]]
    for thr in (2, 3, 4):
        _eq(ctx.common(dl, thr, cts[:4]), O.common(dl, thr, _expand(dl, cts[:4]), tax), ("common dup", thr))
    # union of unsorted files with duplicates (hash-map semantics) and one taxid per file
    sh = [rng.permutation(np.concatenate([f, f[:100]])) for f in files[:4]]
    _eq(ctx.union(sh, cts[:4]), O.union(sh, _expand(sh, cts[:4]), tax))


@pytest.mark.parametrize("knob,value,route", [("UKM_PUNION", "2", 3), ("UKM_PUNION", "1", 3), ("UKM_SRMERGE", "1", 4), ("UKM_KWAY", "1", 2),
                                              ("UKM_NO_KWAY", "1", 1), ("UKM_PUNION_RANKED", "0", 3)])
def test_union_file_taxids_every_route(env, monkeypatch, knob, value, route):
    """the n-file union with one taxid per file through the hash probes (the file's taxid and pre-order number as scalars),
    the single-pass merge, the k-way merge and the pairwise tree (the arrays are built on the device for the merges)"""
    O, L, ctx, tax, pool, kind = env
    monkeypatch.setenv(knob, value)
    if knob == "UKM_PUNION_RANKED":
        monkeypatch.setenv("UKM_PUNION", "2")    # (the generic taxid tables with the files' taxids as scalars)
    elif knob != "UKM_PUNION":
        monkeypatch.setenv("UKM_PUNION", "0")
    rng = np.random.default_rng(3)
    for nfiles, nu, p in ((40, 40_000, 0.5), (600, 9_000, 0.3), (30, 3_000, 0.6)):
        if knob == "UKM_SRMERGE" and nfiles < 600:
            continue
        files = _files(nfiles, nu, p, seed=31)
        for name, taxs in _tax_shapes(files, pool, rng):
            if name == "same":
                continue
            gk, gt = ctx.union(files, taxs)
            if name != "mixed" or not knob.startswith("UKM_PUNION"):
                assert ctx.last_route() == route, (knob, nfiles, name, ctx.last_route())
            _eq((gk, gt), O.union(files, _expand(files, taxs), tax), (knob, nfiles, name))


def test_probe_paths_file_taxids_new_codes_and_aliases(env, monkeypatch):
    """the hash-probe union / counting probes with one taxid per file where later files bring codes the base set lacks
    (claimed in the tables by whichever wave comes first) and the files' taxids are aliases of one another (a merged id
    and its target, two unknown ids: same pre-order number, different taxid) -- the claim race of the round-4 advice: the
    result must be the LCA contract's value whatever wave claimed the code"""
    O, L, ctx, tax, pool, kind = env
    monkeypatch.setenv("UKM_PUNION", "2")
    if kind == "forest":
        alias = [7, 4, 4, 7, 1000, 1001, 7, 4, 6, 0]   # 7 is merged into 4; 1000 / 1001 are unknown
    else:
        alias = [int(pool[0]), int(pool[0]), int(pool[1]), int(pool[0]), 5000000, 5000001, int(pool[1]), int(pool[0]), 0, int(pool[0])]
    U = _universe(30_000, 20)
    rng = np.random.default_rng(12)
    for rep in range(6):
        nfiles = 36
        base = [U[_member(len(U), f, 0.5, 40 + rep)][: 6000] for f in range(8)]       # the largest files: the base set
        fresh = U[_member(len(U), 99, 0.4, 41 + rep)][7000:]                          # codes no base file holds
        later = [np.sort(np.unique(np.concatenate([U[_member(len(U), f, 0.5, 40 + rep)][:5000], fresh[_member(len(fresh), f, 0.7, 9)]])))
                 for f in range(8, nfiles)]
        files = base + later
        taxs = [alias[(i + rep) % len(alias)] for i in range(nfiles)]
        ex = _expand(files, taxs)
        gk, gt = ctx.union(files, taxs)
        assert ctx.last_route() == 3
        _eq((gk, gt), O.union(files, ex, tax), ("union alias", rep))
        _eq(ctx.union(files, ex), O.union(files, ex, tax), ("union alias arrays", rep))
        for thr in (2, nfiles // 2):
            gk, gt = ctx.common(files, thr, taxs)
            _eq((gk, gt), O.common(files, thr, ex, tax), ("common alias", rep, thr))
        gk, gt = ctx.merge_k(files, taxs, mode=L.REPEATED)
        _eq((gk, gt), O.merge_k(files, ex, mode=O.REPEATED, tax=tax), ("merge -d alias", rep))


@pytest.mark.parametrize("clade", ["0", "1", None])
def test_probe_tables_clade_mode_per_record_taxids(env, monkeypatch, clade):
    """Round 5: the probe union's / counting probes' tables with per-record taxids in CLADE MODE (ukm_punion.hip: a record
    brings the one-byte clade code of its taxid; its 4-byte pre-order number is fetched only while the entry's interval
    lies inside one clade) -- UKM_PUNION_CLADE = 1 forces it, 0 forbids it, unset = the sample decides.  Against the oracle
    on files whose later ones bring new codes (claims), with taxids that are unrelated (a hash of the code over the whole
    pool: zeros, unknown and merged ids in the forest), related (drawn from one small clade: every hit needs its number),
    equal within a file, and mixed with files that carry ONE taxid; `union`, `common` below the number of files, `merge -d`."""
    O, L, ctx, tax, pool, kind = env
    monkeypatch.setenv("UKM_PUNION", "2")
    if clade is None:
        monkeypatch.delenv("UKM_PUNION_CLADE", raising=False)
    else:
        monkeypatch.setenv("UKM_PUNION_CLADE", clade)
    U = _universe(40_000, 20)
    nfiles = 30
    base = [U[_member(len(U), f, 0.5, 140)][:8000] for f in range(8)]
    fresh = U[_member(len(U), 99, 0.4, 141)][9000:]
    later = [np.sort(np.unique(np.concatenate([U[_member(len(U), f, 0.5, 140)][:7000], fresh[_member(len(fresh), f, 0.6, 19)]])))
             for f in range(8, nfiles)]
    files = base + later
    related = pool[-6:] if kind == "tree" else np.array([4, 5, 6, 7], dtype=np.uint32)   # (one small clade)
    shapes = {
        "unrelated": [_taxids(f, pool, i) for i, f in enumerate(files)],
        "related": [_taxids(f, related, i) for i, f in enumerate(files)],
        "one_per_file_arrays": [np.full(len(f), int(pool[(3 * i + 1) % len(pool)]), np.uint32) for i, f in enumerate(files)],
        "mixed": [(_taxids(f, pool, i) if i % 3 else int(pool[(5 * i + 2) % len(pool)])) for i, f in enumerate(files)],
    }
    for name, taxs in shapes.items():
        ex = _expand(files, taxs)
        gk, gt = ctx.union(files, taxs)
        assert ctx.last_route() == 3, name
        _eq((gk, gt), O.union(files, ex, tax), ("union", name, clade))
        for thr in (2, nfiles // 2, nfiles - 1):
            _eq(ctx.common(files, thr, taxs), O.common(files, thr, ex, tax), ("common", name, thr, clade))
        _eq(ctx.merge_k(files, taxs, mode=L.REPEATED), O.merge_k(files, ex, mode=O.REPEATED, tax=tax), ("merge -d", name, clade))


def test_clade_folds_on_a_skewed_taxonomy(monkeypatch):
    """The one-byte clade codes are a CUT of the forest that follows its shape (ukm_tax.hip: the clade node with the most ids
    below it is replaced by all of its children while 255 fit), so clade nodes sit at different depths.  The folds that keep
    `code << 24 | number` (probe tables in clade mode, the single pass's emit, the many-file inter) rely on the codes being
    monotone in the pre-order numbers and on different codes implying the LCA of the clade nodes: all three against the oracle
    on a taxonomy with one large kingdom, small ones, a node with 300 children and a chain, taxids over all of it."""
    from oracle import oracle as O
    from unikmer_amd import lib as L
    child, parent = [], []
    nxt = [1]

    def new(par=None):
        t = nxt[0]; nxt[0] += 1
        child.append(t); parent.append(t if par is None else par)
        return t

    def subtree(par, fan, depth):
        if depth:
            for _ in range(fan):
                subtree(new(par), fan, depth - 1)

    r = new()
    big = new(r)
    for _ in range(12):
        subtree(new(big), 7, 3)
    subtree(new(r), 3, 2)
    wide = new(r)
    for _ in range(300):
        new(wide)
    x = new(r)
    for _ in range(30):
        x = new(x)
    child, parent = np.array(child, dtype=np.uint32), np.array(parent, dtype=np.uint32)
    T = int(child.max())
    ctx = L.Context(0)
    ctx.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    ids = np.arange(0, T + 3, dtype=np.uint32)      # (0 and two unknown ids among them)
    U = _universe(40_000, 20)
    nfiles = 40
    files = [U[_member(len(U), f, 0.5, 240)] for f in range(nfiles)]
    taxs = [_taxids(f, ids, i) for i, f in enumerate(files)]
    monkeypatch.setenv("UKM_PUNION", "2")
    monkeypatch.setenv("UKM_PUNION_CLADE", "1")
    _eq(ctx.union(files, taxs), O.union(files, taxs, tax), "probe union, clade mode")
    assert ctx.last_route() == 3
    _eq(ctx.common(files, nfiles // 2, taxs), O.common(files, nfiles // 2, taxs, tax), "counting probes, clade mode")
    monkeypatch.setenv("UKM_PUNION", "0")
    monkeypatch.setenv("UKM_SRMERGE", "1")
    monkeypatch.setenv("UKM_SRMERGE_CLADE", "1")
    _eq(ctx.union(files, taxs), O.union(files, taxs, tax), "single pass, clade emit")
    assert ctx.last_route() == 4
    monkeypatch.delenv("UKM_SRMERGE", raising=False)
    core = _member(len(U), 0, 0.3, 277)
    cfiles = [U[core | _member(len(U), f + 1, 0.6, 278)] for f in range(nfiles)]
    ctaxs = [_taxids(f, ids[1:T + 1], i) for i, f in enumerate(cfiles)]
    exp = O.inter(cfiles, ctaxs, tax)
    assert len(exp[0]) > 1000
    _eq(ctx.inter(cfiles, ctaxs), exp, "many-file inter")
    a, b = cfiles[0], cfiles[1]
    _eq(ctx.setop2(L.OP_UNION, a, b, ctaxs[0], ctaxs[1]), O.union([a, b], [ctaxs[0], ctaxs[1]], tax), "2-way union")
    _eq(ctx.setop2(L.OP_INTER, a, b, ctaxs[0], ctaxs[1]), O.inter([a, b], [ctaxs[0], ctaxs[1]], tax), "2-way inter")
    ctx.close()


def test_inter_diff_common_file_taxids_many_files(env, monkeypatch):
    """1000 files with one taxid each: `inter`, `diff`, `diff -t` and `common` of all files are the PLAIN operation and a
    fill (the probe fold, the chained fold and the synchronous fold in turn); results against the oracle's file-by-file
    loops over the expanded arrays"""
    O, L, ctx, tax, pool, kind = env
    U = _universe(6_000, 20)
    core = _member(len(U), 0, 0.3, 77)
    files = [U[core | _member(len(U), f + 1, 0.6, 78)] for f in range(1000)]
    taxs = [int(pool[(7 * i + 2) % len(pool)]) for i in range(1000)]
    ex = _expand(files, taxs)
    exp_i = O.inter(files, ex, tax)
    exp_d = O.diff(files[:40], ex[:40], tax)
    exp_dt = O.diff(files[:40], ex[:40], tax, compare_taxid=True)
    exp_c = O.common(files[:200], 200, ex[:200], tax)
    for no_pf, no_fold in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("UKM_NO_PFOLD", no_pf)
        monkeypatch.setenv("UKM_NO_FOLD", no_fold)
        _eq(ctx.inter(files, taxs), exp_i, ("inter", no_pf, no_fold))
        _eq(ctx.diff(files[:40], taxs[:40]), exp_d, ("diff", no_pf, no_fold))
        _eq(ctx.diff(files[:40], taxs[:40], compare_taxid=True), exp_dt, ("diff -t", no_pf, no_fold))
        _eq(ctx.common(files[:200], 200, taxs[:200]), exp_c, ("common", no_pf, no_fold))
    assert len(exp_i[0]) > 100 and len(exp_c[0]) > 100


def test_merge_by_placement_with_file_taxids(env, monkeypatch):
    """keep-everything `merge` (mergeChunksFile, util-sort.go:196-225,289-351) of files with ONE taxid each through the
    placement route: the kernel writes the file's taxid as a scalar (no array is built); files that all carry the same taxid
    -- the chunk files of `sort -m` over a `count -t` file -- are the plain merge and a fill"""
    O, L, ctx, tax, pool, kind = env
    monkeypatch.setenv("UKM_PLACE", "1")
    files = _files(120, 9_000, 0.8, seed=3)
    per_file = [int(pool[(5 * i + 2) % len(pool)]) for i in range(len(files))]
    mixed = [per_file[i] if i % 3 else _taxids(f, pool, i) for i, f in enumerate(files)]
    for name, taxs in (("per file", per_file), ("mixed", mixed), ("same", [int(pool[3])] * len(files))):
        ex = _expand(files, taxs)
        for mode in (L.PLAIN, L.UNIQUE, L.REPEATED):
            gk, gt = ctx.merge_k(files, taxs, mode=mode)
            if mode == L.PLAIN and name != "same":
                assert ctx.last_route() == 7, (name, ctx.last_route())
            ok, ot = O.merge_k(files, ex, mode=mode, tax=tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (name, mode)


def test_ranked_probe_union_step_edges(env, monkeypatch):
    """The ranked probe pass (pr_probe_kernel, one taxid per file) with the round-6 streaming skeleton.  A base set of 4e5
    codes gives ranges of ~780 entries (pr_range_for), so later files that hold a half / a third / a twentieth / a thousandth of
    the universe have slices of ~390 / 260 / 39 / 0-2 records: general first steps, full steps, partial and one-record tails,
    slices at either end of their files; plus files of 0, 1 and 2 records and private codes -- against the oracle fed the
    expanded arrays.  Then ONE swapped neighbouring pair anywhere in a later file (inside a lane's pair, between lanes,
    between steps, across a range boundary, at either end): the pass must notice (the call then takes another route and still
    gives the oracle's answer: union.go:186-208 does not need sorted files)."""
    O, L, ctx, tax, pool, kind = env
    rng = np.random.default_rng(6062)
    monkeypatch.setenv("UKM_PUNION", "2")
    U = _universe(400_000, gap_bits=24)
    base = [U[_member(len(U), f, 0.9, 3)] for f in range(8)]
    later = [U[_member(len(U), 20 + i, p, 3)] for i, p in enumerate((0.5, 0.5, 0.33, 0.33, 0.05, 0.05, 0.001, 0.001, 0.6, 0.25))]
    later += [np.empty(0, np.uint64), U[77:78].copy(), U[100:102].copy(), np.array([U[-1] + np.uint64(5)], np.uint64),
              np.unique(rng.integers(1, int(U[-1]), 3000).astype(np.uint64)), U[:129].copy(), U[-257:].copy()]
    files = base + later
    taxs = [int(pool[(5 * i + 3) % len(pool)]) for i in range(len(files))]
    _eq(ctx.union(files, taxs), O.union(files, _expand(files, taxs), tax))
    assert ctx.last_route() == 3
    victim = later[2]
    spots = sorted({0, 1, 2, 126, 127, 128, 129, 254, 255, 256, 257, len(victim) - 2, len(victim) - 3} |
                   set(int(x) for x in rng.integers(0, len(victim) - 1, 14)))
    small = base + [later[4], None, later[6]]
    tt = taxs[:8] + [taxs[12], taxs[10], taxs[14]]
    for sp in spots:
        v = victim.copy()
        v[sp], v[sp + 1] = v[sp + 1], v[sp]
        trial = list(small)
        trial[9] = v
        _eq(ctx.union(trial, tt), O.union(trial, _expand(trial, tt), tax), sp)
        assert ctx.last_route() != 3, sp
    trial = list(small)
    trial[9] = victim
    _eq(ctx.union(trial, tt), O.union(trial, _expand(trial, tt), tax))
    assert ctx.last_route() == 3
