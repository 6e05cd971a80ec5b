"""BASELINE.json's configurations at their FULL sizes on one MI355X (round-2 review: the full-size runs were only
checked for sortedness).  The CPU oracle cannot process 1e9..1e10 records in test time, so each test combines

  * size-independent identities over the WHOLE output (inclusion-exclusion, XOR checksum of checksums, strict order,
    rank consistency between the union and the intersection),
  * bit-exact comparison of 1e6-record WINDOWS cut from both ends and the middle of the outputs with the oracle run on
    the matching slices of the inputs, and
  * equality with an independent device computation of the same result (torch boolean algebra on the generator's
    membership bits; a chain of 2-way kernels against the k-way kernel; the prefix-XOR window kernel against the rolling
    strip kernel).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

W = 1_000_000  # records per oracle-checked window


@pytest.fixture(scope="module")
def env():
    import torch
    import bench
    from unikmer_amd import lib
    from oracle import oracle
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    return torch, bench, lib, ctx, oracle, dev


def _xor(torch, t):
    x = t
    while x.numel() > 1:
        if x.numel() & 1:
            x = torch.cat([x, torch.zeros(1, dtype=x.dtype, device=x.device)])
        h = x.numel() // 2
        x = x[:h] ^ x[h:]
    return int(x.item()) if x.numel() else 0


def _strict(t):
    return bool((t[1:] > t[:-1]).all())


def _np(t):
    return t.cpu().numpy().view(np.uint64)


def _lower(torch, S, v):
    """number of elements of the sorted int64 tensor S (codes < 2^63) below the python int v"""
    return int(torch.searchsorted(S, torch.tensor([v], dtype=torch.int64, device=S.device)).item())


def test_metric_config_2x1e9_full_size(env):
    """BASELINE metric: union + inter (+ diff) of two sorted k=31 sets of 1e9 codes, bench.py's generator."""
    torch, bench, lib, ctx, O, dev = env
    n = 1_000_000_000
    A, B = bench.gen_sets_device((4 * n + 2) // 3, 32, 0, bench.SEED, dev)
    na, nb = A.numel(), B.numel()
    out_u = torch.empty(na + nb, dtype=torch.int64, device=dev)
    out_i = torch.empty(min(na, nb), dtype=torch.int64, device=dev)
    U = ctx.setop2(lib.OP_UNION, A, B, out=out_u)
    I = ctx.setop2(lib.OP_INTER, A, B, out=out_i)
    nu, ni = U.numel(), I.numel()
    assert nu + ni == na + nb                                   # inclusion-exclusion
    assert _strict(U) and _strict(I)
    xa, xb, xu, xi = (_xor(torch, t) for t in (A, B, U, I))
    assert xu == xa ^ xb ^ xi                                   # checksum of checksums
    # windows at both ends and in the middle: the oracle on the matching input slices, and the window's RANK in the
    # output from the inputs (|{u in U: u < v}| = |{a < v}| + |{b < v}| - |{i in I: i < v}|)
    for start in (0, nu // 2 - W // 2, nu - W):
        win = U[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
        b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
        assert np.array_equal(_np(win), O.union([_np(A[a0:a1]), _np(B[b0:b1])])), start
        assert start == a0 + b0 - _lower(torch, I, lo), start
    for start in (0, ni // 2 - W // 2, ni - W):
        win = I[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
        b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
        assert np.array_equal(_np(win), O.inter([_np(A[a0:a1]), _np(B[b0:b1])])), start
        # every element of I below the window is an element of A and of B below it, and U accounts for the rest
        assert start == a0 + b0 - _lower(torch, U, lo), start
    # diff reuses the union buffer
    del U
    D = ctx.setop2(lib.OP_DIFF, A, B, out=out_u)
    nd = D.numel()
    assert nd == na - ni and _strict(D)
    assert _xor(torch, D) == xa ^ xi
    for start in (0, nd // 2 - W // 2, nd - W):
        win = D[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
        b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
        assert np.array_equal(_np(win), O.diff([_np(A[a0:a1]), _np(B[b0:b1])])), start
        assert start == a0 - _lower(torch, I, lo), start


def test_metric_config_2x1e9_with_per_record_taxids_full_size(env, monkeypatch):
    """north_star's second half -- "with per-k-mer TaxId LCA reduction" -- on the metric's two sets at FULL size: every record
    carries a uniformly random taxid of the complete 8-ary tree of depth 7, union and inter through the 2-way
    kernel's taxid instantiation (round 6: small tiles, the LCAs of a tile walked densely behind the merge loop, relatives
    through the fix-up list).  Checked over the WHOLE output: the codes are the plain kernel's; every taxid equals what the
    bulk LCA entry point (ukm_lca: the root-path / clade-table walk, no LDS table, no queue, no fix-up list) gives on the
    taxids of the input records found by torch.searchsorted -- A's own, B's own, or the LCA where both hold the code; inter
    through the two-launch source-word route gives the same arrays; and 1e6-record windows at both ends and in the middle
    against the oracle's ancestor walk."""
    torch, bench, lib, ctx, O, dev = env
    from conftest import synth_tree
    n = 1_000_000_000
    A, B = bench.gen_sets_device((4 * n + 2) // 3, 32, 0, bench.SEED, dev)
    na, nb = A.numel(), B.numel()
    child, parent = synth_tree(7, 8)
    ctx.taxonomy_load(child, parent)
    tax, T = O.Taxonomy(child, parent), len(child)
    ta = (1 + (bench.splitmix64_torch(A ^ bench._i64(bench.SEED + 2)) & ((1 << 40) - 1)) % T).to(torch.int32)
    tb = (1 + (bench.splitmix64_torch(B ^ bench._i64(bench.SEED + 3)) & ((1 << 40) - 1)) % T).to(torch.int32)
    out_k = torch.empty(na + nb, dtype=torch.int64, device=dev)
    out_t = torch.empty(na + nb, dtype=torch.int32, device=dev)
    plain = torch.empty(na + nb, dtype=torch.int64, device=dev)

    def check(op, name):
        K, Tx = ctx.setop2(op, A, B, ta, tb, out=out_k, out_taxids=out_t)
        P = ctx.setop2(op, A, B, out=plain)
        assert K.numel() == P.numel() and bool((K == P).all()), name
        # where every output record comes from
        ia = torch.searchsorted(A, K).clamp_(max=na - 1)
        in_a = A[ia] == K
        ib = torch.searchsorted(B, K).clamp_(max=nb - 1)
        in_b = B[ib] == K
        assert bool((in_a | in_b).all()), name
        va, vb = ta[ia], tb[ib]
        del ia, ib
        both = in_a & in_b
        exp = torch.where(in_a, va, vb)
        idx = both.nonzero().squeeze(1)
        del both
        step = 1 << 28                                            # (bounds the bulk call's temporaries)
        for lo in range(0, idx.numel(), step):
            sel = idx[lo:lo + step]
            exp[sel] = ctx.lca(va[sel].contiguous(), vb[sel].contiguous()).to(torch.int32)
        assert bool((Tx == exp).all()), name
        nmatch = idx.numel()
        del va, vb, exp, idx, in_a, in_b
        # windows against the oracle
        nk = K.numel()
        ofn = O.union if op == lib.OP_UNION else O.inter
        for start in (0, nk // 2 - W // 2, nk - W):
            lo, hi = int(K[start].item()), int(K[start + W - 1].item())
            a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
            b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
            ek, et = ofn([_np(A[a0:a1]), _np(B[b0:b1])], [ta[a0:a1].cpu().numpy().view(np.uint32), tb[b0:b1].cpu().numpy().view(np.uint32)], tax)
            assert np.array_equal(_np(K[start:start + W]), ek) and np.array_equal(Tx[start:start + W].cpu().numpy().view(np.uint32), et), (name, start)
        return nmatch

    m_u = check(lib.OP_UNION, "union")
    m_i = check(lib.OP_INTER, "inter")
    assert m_u == m_i and m_i > 600_000_000                       # every record of the intersection is a matched pair
    # the two-launch source-word route of inter: the same arrays
    ni = m_i
    ref_t = out_t[:ni].clone()
    monkeypatch.setenv("UKM_SETOP_SRC", "1")
    K2, T2 = ctx.setop2(lib.OP_INTER, A, B, ta, tb, out=out_k, out_taxids=out_t)
    monkeypatch.delenv("UKM_SETOP_SRC", raising=False)
    assert K2.numel() == ni and bool((T2 == ref_t).all())


def test_config3_union_of_100_files_x_1e8_full_size(env, monkeypatch):
    """BASELINE config 3 on one GPU: 100 sorted files of ~1e8 codes drawn (p = 0.5) from one universe of 2e8.
    The union — the library's own choice (the hash-probe pass of ukm_punion.hip) AND the k-way streaming merge alone
    (UKM_PUNION=0) — must equal (a) the universe elements that are in at least one file, computed with torch from the
    generator's membership bits, and (b) a chain of 99 two-way unions through the tile kernel."""
    torch, bench, lib, ctx, O, dev = env
    nfiles, per = 100, 100_000_000
    nu = 2 * per
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    Uv = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1)), 0)
    files = []
    anym = torch.zeros(nu, dtype=torch.bool, device=dev)
    for f in range(nfiles):
        m = (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1))) & 1) == 1
        files.append(Uv[m])
        anym |= m
    del j, m
    total = sum(x.numel() for x in files)
    assert 0.99e10 < total < 1.01e10
    expect = Uv[anym]
    del anym
    out = torch.empty(nu + 8, dtype=torch.int64, device=dev)
    monkeypatch.setenv("UKM_PUNION", "0")
    got = ctx.union(files, out=out)
    assert got.numel() == expect.numel() and _strict(got)
    assert bool((got == expect).all())
    monkeypatch.delenv("UKM_PUNION")
    out[:8].zero_()
    got = ctx.union(files, out=out)
    assert got.numel() == expect.numel() and _strict(got)
    assert bool((got == expect).all())
    x = _xor(torch, got)
    del expect
    # oracle on windows: the oracle's hash-map union of the 100 matching slices
    for start in (0, got.numel() // 2 - W // 2, got.numel() - W):
        win = got[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        sl = []
        for f in files:
            s0, s1 = _lower(torch, f, lo), _lower(torch, f, hi + 1)
            sl.append(_np(f[s0:s1]))
        assert np.array_equal(_np(win), O.union(sl)), start
    # the same stream from the 2-way tile kernel, chained (ping-pong buffers)
    buf = [torch.empty(nu + 8, dtype=torch.int64, device=dev) for _ in range(2)]
    acc = files[0]
    for t, f in enumerate(files[1:]):
        acc = ctx.setop2(lib.OP_UNION, acc, f, out=buf[t & 1])
    assert acc.numel() == got.numel() and _xor(torch, acc) == x and bool((acc == got).all())


def test_config3_union_with_taxids_full_size(env, monkeypatch):
    """Config 3's files WITH taxids (union.go:195-201: the TaxId of a code is the LCA over every record that carries it),
    100 x 1e8 records through the hash-probe pass with the TaxId fold in its tables (ukm_punion.hip, route 3).
    (a) every record of file f carries that file's taxid (a leaf of the complete 8-ary tree): the expected TaxId of every
    code is computed with torch from the membership bits — the smallest and the largest leaf among the files that hold
    the code, climbed to their common ancestor by the tree's arithmetic; (b) uniformly random taxids: three windows of the
    output against the oracle's hash-map union of the matching slices of all 100 files."""
    torch, bench, lib, ctx, O, dev = env
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import synth_tree
    child, parent = synth_tree(7, 8)
    ctx.taxonomy_load(child, parent)
    ctx.trim()   # (the plain test's k-way merge left 160 GB of workspace with the context)
    torch.cuda.empty_cache()
    tax = O.Taxonomy(child, parent)
    T = len(child)
    leaves0 = T - 8 ** 7 + 1
    nfiles, per = 100, 100_000_000
    nu = 2 * per
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    Uv = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1)), 0)
    files, taxs = [], []
    anym = torch.zeros(nu, dtype=torch.bool, device=dev)
    mn = torch.full((nu,), 1 << 40, dtype=torch.int64, device=dev)
    mx = torch.zeros(nu, dtype=torch.int64, device=dev)
    for f in range(nfiles):
        m = (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1))) & 1) == 1
        k = Uv[m]
        files.append(k)
        tf = leaves0 + (f * 7919) % (8 ** 7)
        taxs.append(torch.full((k.numel(),), tf, dtype=torch.int32, device=dev))
        anym |= m
        mn = torch.where(m & (mn > tf), torch.full_like(mn, tf), mn)
        mx = torch.where(m & (mx < tf), torch.full_like(mx, tf), mx)
    del j, m
    a, b = mn[anym], mx[anym]
    del mn, mx
    for _ in range(7):   # leaves of one depth: climb both until they meet (parent(t) = (t - 2) // 8 + 1)
        ne = a != b
        a = torch.where(ne, (a - 2) // 8 + 1, a)
        b = torch.where(ne, (b - 2) // 8 + 1, b)
    assert bool((a == b).all())
    expect_k, expect_t = Uv[anym], a.to(torch.int32)
    del anym, b, Uv
    out = torch.empty(nu + 8, dtype=torch.int64, device=dev)
    outt = torch.empty(nu + 8, dtype=torch.int32, device=dev)
    gk, gt = ctx.union(files, taxs, out=out, out_taxids=outt)
    assert ctx.last_route() == 3
    assert gk.numel() == expect_k.numel() and _strict(gk)
    assert bool((gk == expect_k).all()) and bool((gt == expect_t).all())
    del expect_k, expect_t, taxs
    torch.cuda.empty_cache()
    # (b) random taxids
    taxs = [(1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32)
            for f, k in enumerate(files)]
    gk, gt = ctx.union(files, taxs, out=out, out_taxids=outt)
    assert ctx.last_route() == 3 and _strict(gk)
    Ws = 200_000
    for start in (0, gk.numel() // 2 - Ws // 2, gk.numel() - Ws):
        wk, wt = gk[start:start + Ws], gt[start:start + Ws]
        lo, hi = int(wk[0].item()), int(wk[-1].item())
        sl, tl = [], []
        for f, t in zip(files, taxs):
            s0, s1 = _lower(torch, f, lo), _lower(torch, f, hi + 1)
            sl.append(_np(f[s0:s1]))
            tl.append(t[s0:s1].cpu().numpy().view(np.uint32))
        ok, ot = O.union(sl, tl, tax)
        assert np.array_equal(_np(wk), ok) and np.array_equal(wt.cpu().numpy().view(np.uint32), ot), start


def test_config5_sketch_1e10_bases_full_size(env, monkeypatch):
    """BASELINE config 5: ntHash Scaled-MinHash sketch, k = 51, scale 1000, 1e10 bases of 150-bp reads.
    The rolling strip kernel against the prefix-XOR kernel over all 6.7e9 windows (two algorithms), the oracle on the
    reads at both ends, the expected keep rate, and the sort + unique of the sketch."""
    torch, bench, lib, ctx, O, dev = env
    nb = 10_000_000_000
    nb -= nb % 150
    chunk = 1 << 30
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    bases = torch.empty(nb, dtype=torch.uint8, device=dev)
    for lo in range(0, nb, chunk):
        hi = min(lo + chunk, nb)
        i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
        bases[lo:hi] = lut[(w >> (2 * (i & 31))) & 3]
        del i, w
    bases[5_000_000_025:5_000_000_125] = ord("N")                   # a stretch of N inside the middle reads
    reads = torch.arange(0, nb + 1, 150, dtype=torch.int64, device=dev)
    k = 51
    windows = (nb // 150) * (150 - k + 1)
    mh = ctx.max_hash(1000)
    cap = nb // 400
    out_a = torch.empty(cap, dtype=torch.int64, device=dev)
    out_b = torch.empty(cap, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    monkeypatch.setenv("UKM_NTHASH_STRIP", "1")
    a = ctx.nthash(bases, reads, k, canonical=True, max_hash=mh, out=out_a)
    monkeypatch.setenv("UKM_NTHASH_STRIP", "0")
    b = ctx.nthash(bases, reads, k, canonical=True, max_hash=mh, out=out_b)
    monkeypatch.delenv("UKM_NTHASH_STRIP", raising=False)
    assert a.numel() == b.numel() and bool((a == b).all())
    # canonical = min of two (nearly) independent uniform hashes: P(keep) = 1 - (1 - 1/scale)^2
    p = 1.0 - (1.0 - 1.0 / 1000) ** 2
    assert abs(a.numel() - windows * p) < 6 * (windows * p) ** 0.5 + 1e-5 * windows
    assert bool((a >= 0).all()) and int(a.max().item()) <= mh        # maxHash < 2^63
    # oracle on the first and the last 40000 reads (window order: a prefix / suffix of the sketch)
    R = 40_000
    head = O.count_windows(bases[:R * 150].cpu().numpy(), np.arange(0, R * 150 + 1, 150, dtype=np.uint64), k,
                           hashed=True, canonical=True, max_hash=mh)
    tail = O.count_windows(bases[nb - R * 150:].cpu().numpy(), np.arange(0, R * 150 + 1, 150, dtype=np.uint64), k,
                           hashed=True, canonical=True, max_hash=mh)
    assert len(head) > 5000 and np.array_equal(_np(a[:len(head)]), head)
    assert len(tail) > 5000 and np.array_equal(_np(a[a.numel() - len(tail):]), tail)
    # ... and on 40000 reads around the N stretch in the middle, located in the sketch by its first hash
    m0 = (5_000_000_025 // 150 - R // 2) * 150
    mid = O.count_windows(bases[m0:m0 + R * 150].cpu().numpy(), np.arange(0, R * 150 + 1, 150, dtype=np.uint64), k,
                          hashed=True, canonical=True, max_hash=mh)
    pos = torch.nonzero(a == int(mid[0])).flatten()                  # kept hashes are <= maxHash < 2^63
    assert pos.numel() >= 1
    assert any(np.array_equal(_np(a[int(q):int(q) + len(mid)]), mid) for q in pos.tolist())
    # the sketch as a set: sort + unique (the count path's tail)
    xs = _xor(torch, a)
    n_kept = a.numel()
    ctx.sort_u64(a, int(mh).bit_length())
    assert bool((a[1:] >= a[:-1]).all()) and _xor(torch, a) == xs
    u = ctx.unique(a, out=out_b)
    assert _strict(u) and 0.99 * n_kept < u.numel() <= n_kept


def _synth_bases(torch, bench, n, dev):
    """base i = "ACGT"[splitmix64(seed ^ (i >> 5)) >> (2 (i & 31)) & 3]"""
    i = torch.arange(n, dtype=torch.int64, device=dev)
    w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    return lut[(w >> (2 * (i & 31))) & 3]


def test_config2_count_sort_100Mbp_full_size(env, monkeypatch, genomes):
    """BASELINE config 2 end to end at its real size: 100 records x 1 Mbp -> 2-bit encode + canonical (strip kernel) ->
    radix sort (two scatter passes over the top bits, then LDS buckets in their size classes: the route the canonical
    k-mer distribution takes from 2^23 windows on) -> unique, element for element against the oracle's rolling encoder,
    its sort and its scan (count.go:285-436,581; sort.go:463,541-550).  Then a real-genome-like input of 2.5e7 bases
    (the three fixture genomes twice, the second copy with every 97th base changed): low-complexity k-mers crowd single
    buckets of the top-bits route.  Both also with the route switched off (all passes through HBM)."""
    torch, bench, lib, ctx, O, dev = env
    nb = 100_000_000
    bases = _synth_bases(torch, bench, nb, dev)
    off = np.array([nb * r // 100 for r in range(101)], dtype=np.uint64)
    hb = bases.cpu().numpy()
    ow = O.count_windows(hb, off, 31)
    assert len(ow) == nb - 100 * 30
    osorted = O.sort_u64(ow)
    ou = O.unique(osorted)
    doff = torch.from_numpy(off.view(np.int64)).to(dev)
    for knob in (None, "0"):
        if knob is None:
            monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
        else:
            monkeypatch.setenv("UKM_SORT_LOCAL", knob)
        c = ctx.encode_kmers(bases, doff, 31, canonical=True)
        assert np.array_equal(_np(c), ow), knob                 # window order, every code
        ctx.sort_u64(c, 62)
        assert np.array_equal(_np(c), osorted), knob
        u = ctx.unique(c)
        assert np.array_equal(_np(u), ou), knob
        del c, u
        # the same through the one-call entry point (ukm_count): on the bucket route the sort takes its first histogram from
        # the strip encode kernel instead of a pass of its own
        fused0 = ctx.stat("sort_fused_hist")
        assert np.array_equal(_np(ctx.count(bases, doff, 31, canonical=True)), ou), knob
        assert ctx.stat("sort_fused_hist") == fused0 + (1 if knob is None else 0), knob
    monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
    del bases, hb, ow, osorted, ou
    # the fixture genomes, twice
    from conftest import MG1655, IAI39, AMUC
    seqs, offs, at = [], [0], 0
    for rep in range(2):
        for name in (MG1655, IAI39, AMUC):
            s, o = genomes(name)
            s = s.copy()
            if rep:
                s[::97] = np.frombuffer(b"CATG", dtype=np.uint8)[(s[::97] >> 1) & 3]   # a deterministic substitution
            seqs.append(s)
            for b in o[1:]:
                offs.append(at + int(b))
            at += len(s)
    seq = np.concatenate(seqs)
    goff = np.array(offs, dtype=np.uint64)
    gw = O.count_windows(seq, goff, 31)
    assert len(gw) >= 1 << 23                                    # the top-bits route is eligible
    gs = O.sort_u64(gw)
    gu = O.unique(gs)
    dseq = torch.from_numpy(seq).to(dev)
    dgoff = torch.from_numpy(goff.view(np.int64)).to(dev)
    for knob in (None, "0"):
        if knob is None:
            monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
        else:
            monkeypatch.setenv("UKM_SORT_LOCAL", knob)
        c = ctx.encode_kmers(dseq, dgoff, 31, canonical=True)
        assert np.array_equal(_np(c), gw), knob
        ctx.sort_u64(c, 62)
        assert np.array_equal(_np(c), gs), knob
        assert np.array_equal(_np(ctx.unique(c)), gu), knob
        assert np.array_equal(_np(ctx.count(dseq, dgoff, 31, canonical=True)), gu), knob
        for mode, omode in ((lib.REPEATED, O.REPEATED), (lib.SINGLETON, O.SINGLETON)):
            assert np.array_equal(_np(ctx.count(dseq, dgoff, 31, canonical=True, mode=mode)), O.unique(gs, mode=omode)), (knob, mode)
    monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)


def _config4_files(torch, bench, dev, nfiles, per, T, core_share):
    """independent p = 0.9 membership draws over one universe, taxid = 1 + splitmix64(seed3 ^ code)
    mod T per record.  core_share > 0: that share of the universe is in EVERY file and the first file gets private codes,
    so inter / diff / diff -t all fold over all files (without it 0.9^n empties the running result after ~130 files)."""
    nu = int(per / 0.9)
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1)), 0)
    thr = int(0.9 * (1 << 20))
    core = None
    if core_share > 0:
        core = ((bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 77)) >> 11) & ((1 << 20) - 1)) < int(core_share * (1 << 20))
    files, taxs = [], []
    for f in range(nfiles):
        h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
        m = ((h >> 11) & ((1 << 20) - 1)) < thr
        if core is not None:
            m = m | core
        k = U[m]
        if core is not None and f == 0:
            k = torch.cat([k, U[-1] + 1 + torch.arange(per // 10, dtype=torch.int64, device=dev) * 3])
        files.append(k)
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
    return files, taxs


@pytest.mark.parametrize("core_share", [0.3, 0.0])
def test_config4_inter_diff_common_1000_files_x_1e6_full_size(env, monkeypatch, core_share):
    """BASELINE config 4 at its real size -- 1000 files x 1e6 k-mers with per-record taxids on the 8-ary depth-7 tree --
    bit-exact against the oracle's file-by-file loops (inter.go:205-286, diff.go:379-454 incl. -t, common.go:220-344):
    the shape in which every file matters (a 30 % core in every file, private codes in the first) and SURVEY 8(d)'s
    plain p = 0.9 draws (inter and diff run empty after ~130 files: the early exits).  At this size the one-launch folds
    choose range lengths, residency and multi-chunk slices that the small tests never see.  Default routes and
    UKM_NO_PFOLD=1 (the range fold / chained fold behind the hash-probe fold)."""
    torch, bench, lib, ctx, O, dev = env
    from conftest import synth_tree
    child, parent = synth_tree(7, 8)
    ctx.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    nfiles, per = 1000, 1_000_000
    files, taxs = _config4_files(torch, bench, dev, nfiles, per, len(child), core_share)
    assert 0.85e9 < sum(x.numel() for x in files) < 1.3e9
    hf = [_np(x) for x in files]
    ht = [t.cpu().numpy().view(np.uint32) for t in taxs]
    # (the oracle's loops are single-threaded C behind ctypes, which releases the GIL: every operation acts code by code, so
    #  its passes over all 1e9 records run value range by value range on the host's cores -- conftest.oracle_by_value_ranges;
    #  round 5 ran one thread per operation: 137 s of a 556 s suite)
    from conftest import oracle_by_value_ranges as by_ranges
    want = {"inter": by_ranges(lambda k, t: O.inter(k, t, tax), hf, ht, kind="inter"),
            "diff": by_ranges(lambda k, t: O.diff(k, t, tax), hf, ht),
            "diff_t": by_ranges(lambda k, t: O.diff(k, t, tax, compare_taxid=True), hf, ht)}
    if core_share > 0:
        want["common"] = by_ranges(lambda k, t: O.common(k, nfiles, t, tax), hf, ht)
        want["common_minus_1"] = by_ranges(lambda k, t: O.common(k, nfiles - 1, t, tax), hf, ht)
    if core_share > 0:
        assert len(want["inter"][0]) > 200_000 and len(want["diff"][0]) >= per // 10    # results that survive every file
    else:
        assert len(want["inter"][0]) == 0 and len(want["diff"][0]) == 0 and len(want["diff_t"][0]) > 0
    del hf, ht
    cap = files[0].numel() + 8
    ok = torch.empty(cap, dtype=torch.int64, device=dev)
    ot = torch.empty(cap, dtype=torch.int32, device=dev)

    def same(got, name, tag):
        gk, gt = got
        wk, wt = want[name]
        assert gk.numel() == len(wk), (name, tag, gk.numel(), len(wk))
        assert np.array_equal(_np(gk), wk) and np.array_equal(gt.cpu().numpy().view(np.uint32), wt), (name, tag)
    for no_pf in (None, "1"):
        if no_pf is None:
            monkeypatch.delenv("UKM_NO_PFOLD", raising=False)
        else:
            monkeypatch.setenv("UKM_NO_PFOLD", no_pf)
        same(ctx.inter(files, taxs, out=ok, out_taxids=ot), "inter", no_pf)
        same(ctx.diff(files, taxs, out=ok, out_taxids=ot), "diff", no_pf)
        same(ctx.diff(files, taxs, compare_taxid=True, out=ok, out_taxids=ot), "diff_t", no_pf)
    monkeypatch.delenv("UKM_NO_PFOLD", raising=False)
    if core_share > 0:
        # `common` of all files: the probe fold, and the counting merge behind it (keep-everything merge + run scan)
        same(ctx.common(files, nfiles, taxs, out=ok, out_taxids=ot), "common", "probe")
        total = sum(x.numel() for x in files)
        okc = torch.empty(total + 8, dtype=torch.int64, device=dev)
        otc = torch.empty(total + 8, dtype=torch.int32, device=dev)
        monkeypatch.setenv("UKM_COMMON_PROBE", "0")
        same(ctx.common(files, nfiles, taxs, out=okc, out_taxids=otc), "common", "counting probes")
        assert ctx.last_route() == 6   # (hash probes against the first file with a record count per entry, ukm_punion.hip)
        monkeypatch.delenv("UKM_COMMON_PROBE")
        # one below the number of files: codes that one file lacks survive too
        same(ctx.common(files, nfiles - 1, taxs, out=okc, out_taxids=otc), "common_minus_1", "counting probes")
        assert ctx.last_route() == 6
        assert len(want["common_minus_1"][0]) >= len(want["common"][0])
        # ... and through the single-pass merge counting inside its tiles (ukm_srmerge.hip)
        monkeypatch.setenv("UKM_PUNION", "0")
        same(ctx.common(files, nfiles - 1, taxs, out=okc, out_taxids=otc), "common_minus_1", "counted single pass")
        assert ctx.last_route() == 5
        monkeypatch.delenv("UKM_PUNION")


def _draw_files(torch, bench, dev, nfiles, nu, p, T, salt):
    """nfiles independent membership draws (probability p) over ONE universe of nu codes, a per-record taxid each"""
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + salt)) & ((1 << 32) - 1)), 0)
    thr = int(p * (1 << 24))
    files, taxs = [], []
    for f in range(nfiles):
        h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + salt + 1000 * (f + 1)))
        k = U[((h >> 11) & ((1 << 24) - 1)) < thr]
        files.append(k)
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
    return files, taxs


def test_default_routes_1000_files_x_1e6_full_size(env, monkeypatch):
    """The routes the library picks BY DEFAULT for 1000 files x 1e6 records with taxids, at the size where it picks them, each
    element for element against the oracle (round-4 review: they were only compared forced, on universes of <= 50,000 codes):
      * keep-everything `merge` (mergeChunksFile's heap, util-sort.go:196-225,289-351) of files that share their codes
        -> placement (route 7: range lengths, TaxIds eight files at a time, 16-bit code indices);
      * the same of files that hold a fiftieth of a universe each -> the single pass (route 4: by-value passes of ranges
        that do not fit a tile);
      * `union` (union.go:186-305) of files that hold a tenth of a universe each -> the hash-probe union, whose first base
        set misses the hit-rate guard and whose SECOND attempt with four times the files runs (ukm_ctx_get_stat).
    No knob is set.  The oracle's heap merges of 1e9 records run value range by value range on the host's cores."""
    torch, bench, lib, ctx, O, dev = env
    from conftest import synth_tree
    for v in ("UKM_PUNION", "UKM_PLACE", "UKM_SRMERGE", "UKM_KWAY", "UKM_NO_KWAY", "UKM_PUNION_TAX"):
        monkeypatch.delenv(v, raising=False)
    child, parent = synth_tree(7, 8)
    ctx.taxonomy_load(child, parent)
    tax = O.Taxonomy(child, parent)
    T = len(child)
    nfiles, per = 1000, 1_000_000
    shapes = {
        "place": _config4_files(torch, bench, dev, nfiles, per, T, 0.3),                    # 90 % of a universe + a core
        "single": _draw_files(torch, bench, dev, nfiles, 50 * per, 0.02, T, 31),            # a fiftieth of a universe each
        "probe": _draw_files(torch, bench, dev, nfiles, 10 * per, 0.1, T, 47),              # a tenth each
    }
    host = {k: ([_np(x) for x in fs], [t.cpu().numpy().view(np.uint32) for t in ts]) for k, (fs, ts) in shapes.items()}
    from conftest import oracle_by_value_ranges as by_ranges
    # (value range by value range on the host's cores: the heap merge is stable in the file order inside every range)
    want = {
        "place": by_ranges(lambda k, t: O.merge_k(k, t, mode=O.PLAIN, tax=tax), *host["place"]),
        "single": by_ranges(lambda k, t: O.merge_k(k, t, mode=O.PLAIN, tax=tax), *host["single"]),
        "probe": by_ranges(lambda k, t: O.union(k, t, tax), *host["probe"]),
    }
    del host

    def same(got, name):
        gk, gt = got
        wk, wt = want[name]
        assert gk.numel() == len(wk), (name, gk.numel(), len(wk))
        assert np.array_equal(_np(gk), wk), name
        assert np.array_equal(gt.cpu().numpy().view(np.uint32), wt), name
    for name, route in (("place", 7), ("single", 4)):
        files, taxs = shapes[name]
        total = sum(x.numel() for x in files)
        assert 0.8e9 < total < 1.3e9
        ok = torch.empty(total + 8, dtype=torch.int64, device=dev)
        ot = torch.empty(total + 8, dtype=torch.int32, device=dev)
        got = ctx.merge_k(files, taxs, mode=lib.PLAIN, out=ok, out_taxids=ot)
        assert ctx.last_route() == route, (name, ctx.last_route())
        same(got, name)
        del ok, ot, got
    files, taxs = shapes["probe"]
    total = sum(x.numel() for x in files)
    ok = torch.empty(10 * per + 8, dtype=torch.int64, device=dev)
    ot = torch.empty(10 * per + 8, dtype=torch.int32, device=dev)
    got = ctx.union(files, taxs, out=ok, out_taxids=ot)
    assert ctx.last_route() == 3 and ctx.stat("punion_attempts") == 2, (ctx.last_route(), ctx.stat("punion_attempts"))
    same(got, "probe")


def _sorted_in_slices(t, step=1 << 30):
    """non-decreasing, checked slice by slice (bounds torch's temporaries on multi-GB tensors)"""
    n = t.numel()
    return all(bool((t[lo + 1:min(n, lo + step + 1)] >= t[lo:min(n, lo + step + 1) - 1]).all()) for lo in range(0, n - 1, step))


def _sort_windows_match_oracle(torch, O, src, srt, width=1_000_000):
    """windows at both ends and in the middle of the sorted array `srt` against the ORACLE's sort of the records of the
    unsorted array `src` whose values fall into the window's value range (selected with a torch mask)"""
    n = srt.numel()
    for start in (0, n // 2 - width // 2, n - width):
        win = srt[start:start + width]
        lo, hi = int(win[0].item()), int(win[-1].item())
        sel = torch.cat([c[(c >= lo) & (c <= hi)] for c in torch.split(src, 1 << 30)])   # (masks of <= 2^30 elements)
        # (equal values at the window's edges may extend beyond it: compare the window inside the selection)
        want = O.sort_u64(_np(sel))
        below = int((srt[max(0, start - 4096):start] == lo).sum().item()) if start else 0
        assert below < 4096
        assert np.array_equal(_np(win), want[below:below + width]), start


def test_beyond_2_30_and_2_32_records(env, monkeypatch):
    """The paths that only sizes beyond 2^30 / 2^32 reach (round-5 review: checked once by hand in round 1 with
    tools/large_checks.py, while ukm_sort.hip gained the bucket route and its counting step since):
      (1) sortutil.Uint64s (sort.go:463) on more than 2^30 keys -- the radix sort's look-back status words are 64-bit from
          2^30 on -- through the CURRENT top-bits + LDS-bucket route and with UKM_SORT_LOCAL=0 (all passes through HBM);
      (2) union / inter (union.go:186-305, inter.go:205-278) of two sets with |A| + |B| > 2^32 (64-bit tile indices);
      (3) a sort of 2^32 + delta keys: the chunk-and-merge path include/unikmer_hip.h promises from 2^32 records on.
    Each checked over the WHOLE output by order + XOR checksum (+ inclusion-exclusion), on 1e6-record windows against the
    oracle, and (1) element for element against torch.sort (rocPRIM: an independent device sort)."""
    torch, bench, lib, ctx, O, dev = env
    ctx.trim()
    torch.cuda.empty_cache()
    # ---- (1) n > 2^30, 62-bit keys with duplicates
    n = (1 << 30) + 77_777_777
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    K = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    K[12345:12345 + 1000] = K[999]                                   # a run of equal keys
    x0 = _xor(torch, K)
    ref = torch.sort(K).values
    for knob in (None, "0"):
        if knob is None:
            monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
        else:
            monkeypatch.setenv("UKM_SORT_LOCAL", knob)
        S = K.clone()
        ctx.sort_u64(S, 62)
        assert _sorted_in_slices(S) and _xor(torch, S) == x0, knob
        assert torch.equal(S, ref), knob
        if knob is None:
            _sort_windows_match_oracle(torch, O, K, S)
        del S
    monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
    del K, ref
    ctx.trim()
    torch.cuda.empty_cache()
    # ---- (2) |A| + |B| > 2^32
    nu = 3_070_000_000                                               # universe; |A| ~ |B| ~ 0.75 nu
    A, B = bench.gen_sets_device(nu, 29, 0, bench.SEED + 3, dev)
    na, nb = A.numel(), B.numel()
    assert na + nb > (1 << 32)
    out = torch.empty(na + nb, dtype=torch.int64, device=dev)
    xa, xb = _xor(torch, A), _xor(torch, B)
    U = ctx.setop2(lib.OP_UNION, A, B, out=out)
    nuo, xu = U.numel(), _xor(torch, U)
    assert _sorted_in_slices(U) and bool((U[1:1 << 28] > U[:(1 << 28) - 1]).all())
    wins_u = []
    for start in (0, nuo // 2 - W // 2, nuo - W):
        win = U[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
        b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
        assert np.array_equal(_np(win), O.union([_np(A[a0:a1]), _np(B[b0:b1])])), start
        wins_u.append((start, a0, b0, lo))
    del U
    I = ctx.setop2(lib.OP_INTER, A, B, out=out)
    ni, xi = I.numel(), _xor(torch, I)
    assert nuo + ni == na + nb                                       # inclusion-exclusion
    assert xu == xa ^ xb ^ xi                                        # checksum of checksums
    assert _sorted_in_slices(I)
    for start, a0, b0, lo in wins_u:                                 # the union windows' RANK from the inputs and I
        assert start == a0 + b0 - _lower(torch, I, lo), start
    for start in (0, ni // 2 - W // 2, ni - W):
        win = I[start:start + W]
        lo, hi = int(win[0].item()), int(win[-1].item())
        a0, a1 = _lower(torch, A, lo), _lower(torch, A, hi + 1)
        b0, b1 = _lower(torch, B, lo), _lower(torch, B, hi + 1)
        assert np.array_equal(_np(win), O.inter([_np(A[a0:a1]), _np(B[b0:b1])])), start
    del A, B, I, out
    ctx.trim()
    torch.cuda.empty_cache()
    # ---- (3) 2^32 + delta keys
    n = (1 << 32) + 123_456_789
    g.manual_seed(9)
    K = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    x0 = _xor(torch, K)
    S = K.clone()
    ctx.sort_u64(S, 62)
    assert _sorted_in_slices(S) and _xor(torch, S) == x0
    _sort_windows_match_oracle(torch, O, K, S)
    del K, S
    ctx.trim()
    torch.cuda.empty_cache()
