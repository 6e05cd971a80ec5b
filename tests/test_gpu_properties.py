"""Size-independent parity properties at sizes the CPU oracle cannot reach in test time
(SURVEY.md §8(c)/(d)): inclusion-exclusion, checksum-of-checksums, sortedness, idempotence,
permutation invariance.  Inputs come from bench.py's device generator (verified against numpy
there and in test_dist_gloo / bench itself)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N = 50_000_000  # per set


@pytest.fixture(scope="module")
def env():
    import torch
    import bench
    from unikmer_amd import lib
    dev = torch.device("cuda", 0)
    ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    A, B = bench.gen_sets_device((4 * N + 2) // 3, 32, 0, bench.SEED, dev)
    return torch, bench, lib, ctx, A, B


def _xor(torch, t):
    # XOR checksum of an int64 tensor (bit patterns of the uint64 codes)
    x = t
    while x.numel() > 1:
        if x.numel() & 1:
            x = torch.cat([x, torch.zeros(1, dtype=x.dtype, device=x.device)])
        h = x.numel() // 2
        x = x[:h] ^ x[h:]
    return int(x.item()) if x.numel() else 0


def _strict(t):
    return bool((t[1:] > t[:-1]).all())


def test_set_algebra_identities(env):
    torch, bench, lib, ctx, A, B = env
    U = ctx.setop2(lib.OP_UNION, A, B)
    I = ctx.setop2(lib.OP_INTER, A, B)
    D = ctx.setop2(lib.OP_DIFF, A, B)
    D2 = ctx.setop2(lib.OP_DIFF, B, A)
    na, nb = A.numel(), B.numel()
    assert U.numel() + I.numel() == na + nb                      # inclusion-exclusion
    assert D.numel() == na - I.numel() and D2.numel() == nb - I.numel()
    assert U.numel() == D.numel() + D2.numel() + I.numel()
    for t in (U, I, D, D2):
        assert _strict(t)
    xa, xb, xu, xi, xd, xd2 = (_xor(torch, t) for t in (A, B, U, I, D, D2))
    assert xu == xa ^ xb ^ xi                                    # checksum of checksums
    assert xd == xa ^ xi and xd2 == xb ^ xi
    # idempotence / absorption
    assert torch.equal(ctx.setop2(lib.OP_UNION, U, A), U)
    assert torch.equal(ctx.setop2(lib.OP_INTER, U, A), A)
    assert torch.equal(ctx.setop2(lib.OP_INTER, I, A), I)
    assert ctx.setop2(lib.OP_DIFF, I, A).numel() == 0
    assert torch.equal(ctx.setop2(lib.OP_UNION, D, I), A)         # A = (A \\ B) u (A n B)
    # n-way entry points agree with the 2-way kernel
    assert torch.equal(ctx.union([A, B]), U) and torch.equal(ctx.inter([A, B]), I) and torch.equal(ctx.diff([A, B]), D)
    assert torch.equal(ctx.common([A, B], 2), I) and torch.equal(ctx.common([A, B], 1), U)
    assert torch.equal(ctx.merge_k([A, B], mode=lib.UNIQUE), U) and torch.equal(ctx.merge_k([A, B], mode=lib.REPEATED), I)
    M = ctx.merge_k([A, B], mode=lib.PLAIN)
    assert M.numel() == na + nb and bool((M[1:] >= M[:-1]).all()) and _xor(torch, M) == xa ^ xb


def test_sort_is_a_sorted_permutation(env):
    torch, bench, lib, ctx, A, B = env
    g = torch.Generator(device=A.device)
    g.manual_seed(3)
    perm = torch.randperm(A.numel(), device=A.device, generator=g)
    shuffled = A[perm].contiguous()
    work = shuffled.clone()
    ctx.sort_u64(work, 62)
    assert torch.equal(work, A)                                  # A is strictly increasing: the sort must restore it
    # pairs: the payload follows its key
    work = shuffled.clone()
    vals = perm.to(torch.int32).contiguous()
    ctx.sort_pairs(work, vals, 62)
    assert torch.equal(work, A) and torch.equal(vals.long(), torch.arange(A.numel(), device=A.device))
    # multiset: sort(cat(A, B)) then unique == union; repeated == inter; singleton == symmetric difference
    cat = torch.cat([B, A])
    ctx.sort_u64(cat, 62)
    U = ctx.setop2(lib.OP_UNION, A, B)
    I = ctx.setop2(lib.OP_INTER, A, B)
    assert torch.equal(ctx.unique(cat, mode=lib.UNIQUE), U)
    assert torch.equal(ctx.unique(cat, mode=lib.REPEATED), I)
    assert ctx.unique(cat, mode=lib.SINGLETON).numel() == U.numel() - I.numel()
    two = ctx.unique(cat, mode=lib.REPEATED_CHUNK)               # every code once, repeated ones twice
    assert two.numel() == U.numel() + I.numel() and torch.equal(ctx.unique(two, mode=lib.REPEATED), I)
    assert torch.equal(ctx.unique(U, mode=lib.UNIQUE), U)         # idempotent


def test_encode_and_hash_window_counts(env, monkeypatch):
    torch, bench, lib, ctx, A, B = env
    monkeypatch.delenv("UKM_WIN_STRIP", raising=False)
    nb = 40_000_000
    i = torch.arange(nb, dtype=torch.int64, device=A.device)
    w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=A.device)[(w >> (2 * (i & 31))) & 3]
    reads = torch.arange(0, nb + 1, 150, dtype=torch.int64, device=A.device)
    if int(reads[-1]) != nb:
        reads = torch.cat([reads, torch.tensor([nb], dtype=torch.int64, device=A.device)])
    k = 31
    n_rec = reads.numel() - 1
    lens = reads[1:] - reads[:-1]
    expect = int((lens - (k - 1)).clamp(min=0).sum())
    codes = ctx.encode_kmers(bases, reads, k, canonical=False)
    assert codes.numel() == expect and int(codes.max()) < (1 << 62)
    canon = ctx.encode_kmers(bases, reads, k, canonical=True)
    assert bool((canon <= codes).all())                          # canonical = min(fwd, revcomp) as unsigned (< 2^62)
    # neighbouring windows of one read overlap in k-1 bases: code[i+1] == ((code[i] << 2) & mask) | next base
    first = codes[: 150 - k + 1]
    mask = (1 << (2 * k)) - 1
    assert bool((((first[:-1] << 2) & mask) >> 2 == (first[1:] >> 2)).all())
    h = ctx.nthash(bases, reads, k, canonical=True)
    assert h.numel() == expect
    # 150-bp reads: the library's choice (the rolling strip kernel since round 2) against the general kernel
    h51 = ctx.nthash(bases, reads, 51, canonical=False)
    monkeypatch.setenv("UKM_WIN_STRIP", "0")
    assert torch.equal(ctx.encode_kmers(bases, reads, k, canonical=True), canon)
    assert torch.equal(ctx.nthash(bases, reads, k, canonical=True), h)
    assert torch.equal(ctx.nthash(bases, reads, 51, canonical=False), h51)
    monkeypatch.delenv("UKM_WIN_STRIP", raising=False)
    del h51
    mh = ctx.max_hash(100)
    hs = ctx.nthash(bases, reads, k, canonical=True, max_hash=mh)
    # the fused Scaled filter keeps exactly the hashes <= maxHash, in window order
    keep = (h >= 0) & (h <= mh)                                  # maxHash < 2^63, so signed compare on the non-negative half
    assert torch.equal(hs, h[keep])
    frac = hs.numel() / h.numel()
    assert 0.019 < frac < 0.021                                  # canonical = min of two uniform hashes: ~2/scale
    mins = ctx.minimizer(bases, reads, k, 15)
    assert 0 < mins.numel() < h.numel() and bool(torch.isin(mins[:1000], h).all())


def test_sort_pairs_large_is_a_stable_permutation(env):
    """3e7 (code, taxid) records with many equal codes through the fused-histogram radix passes (the next pass's
    digit counts are accumulated while a pass scatters): sorted by code, payloads follow their codes, equal codes
    keep input order (sorts.Quicksort is unstable in the reference; stable is a valid refinement)."""
    torch, bench, lib, ctx, A, B = env
    n = 30_000_000
    g = torch.Generator(device=A.device)
    g.manual_seed(7)
    keys = torch.randint(0, 1 << 22, (n,), dtype=torch.int64, device=A.device, generator=g) << 20
    vals = torch.arange(n, dtype=torch.int32, device=A.device)
    k2, v2 = keys.clone(), vals.clone()
    torch.cuda.synchronize()
    ctx.sort_pairs(k2, v2, 62)
    assert bool((k2[1:] >= k2[:-1]).all())
    assert bool((keys[v2.long()] == k2).all())                      # every payload still sits next to its code
    same = k2[1:] == k2[:-1]
    assert bool((v2[1:][same] > v2[:-1][same]).all())               # stable
    assert _xor(torch, v2.long()) == _xor(torch, vals.long())       # a permutation of the payloads


def test_kway_union_equals_chained_two_way_unions_large(env):
    """12 files x 4e7 codes (4.8e8 records): the k-way streaming union (ukm_union) against a chain of 2-way unions
    through the tile kernel (ukm_setop2) — two independent device paths must give the same stream; the keep-everything
    k-way merge has the same multiset as the inputs (size + XOR checksum) and is sorted."""
    torch, bench, lib, ctx, A, B = env
    dev = A.device
    nu = 80_000_000
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 5)) & ((1 << 30) - 1)), 0)
    files = [U[(bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1))) & 1) == 1] for f in range(12)]
    del j
    torch.cuda.synchronize()
    got = ctx.union(files)
    acc = files[0]
    for f in files[1:]:
        acc = ctx.setop2(lib.OP_UNION, acc, f)
    assert got.numel() == acc.numel() and bool((got == acc).all())
    assert _strict(got)
    del acc, got
    m = ctx.merge_k(files[:6])
    assert m.numel() == sum(f.numel() for f in files[:6])
    assert bool((m[1:] >= m[:-1]).all())
    x = 0
    for f in files[:6]:
        x ^= _xor(torch, f)
    assert _xor(torch, m) == x


def test_nthash_sketch_strip_kernel_equals_general_kernel_large(env, monkeypatch):
    """1e9 bases of 150-bp reads, k = 51, scale 1000: the rolling strip kernel and the prefix-XOR kernel (two
    different algorithms for the same hash) must keep exactly the same windows in the same order."""
    torch, bench, lib, ctx, A, B = env
    dev = A.device
    nb = 1_000_000_050
    i = torch.arange(nb, dtype=torch.int64, device=dev)
    w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
    code = ((w >> (2 * (i & 31))) & 3)
    del i, w
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[code]
    del code
    bases[123_456_789:123_456_889] = ord("N")
    reads = torch.arange(0, nb + 1, 150, dtype=torch.int64, device=dev)
    mh = ctx.max_hash(1000)
    torch.cuda.synchronize()
    monkeypatch.setenv("UKM_NTHASH_STRIP", "1")
    a = ctx.nthash(bases, reads, 51, canonical=True, max_hash=mh).clone()
    monkeypatch.setenv("UKM_NTHASH_STRIP", "0")
    b = ctx.nthash(bases, reads, 51, canonical=True, max_hash=mh)
    assert a.numel() == b.numel() and a.numel() > 1_000_000 and bool((a == b).all())


def test_window_strip_kernel_equals_general_kernel_large(env, monkeypatch):
    """1e9 bases in 37 chromosome-sized records of uneven length: the rolling strip kernel (long records) and the
    prefix-word kernel — two different algorithms — must produce identical codes (k = 31) and hashes (k = 51) for
    every window; neighbouring windows of the codes overlap in k - 1 bases."""
    torch, bench, lib, ctx, A, B = env
    dev = A.device
    nb = 1_000_000_007
    i = torch.arange(nb, dtype=torch.int64, device=dev)
    w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED + 3))
    code = ((w >> (2 * (i & 31))) & 3)
    del i, w
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[code]
    del code
    bases[500_000_000:500_000_200] = ord("N")
    bases[777_777_777] = ord("y")
    cuts = sorted(set([0, nb] + [int(nb * (r * r + 3 * r)) // (37 * 37 + 3 * 37) for r in range(1, 37)] + [123_456_789, 123_456_790, 123_456_800]))
    off = torch.tensor(cuts, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for fn, k in ((ctx.encode_kmers, 31), (ctx.nthash, 51)):
        for canonical in (True, False):
            monkeypatch.setenv("UKM_WIN_STRIP", "1")
            a = fn(bases, off, k, canonical=canonical).clone()
            monkeypatch.setenv("UKM_WIN_STRIP", "0")
            b = fn(bases, off, k, canonical=canonical)
            assert a.numel() == b.numel() and a.numel() > 990_000_000 and bool((a == b).all()), (k, canonical)
            if fn == ctx.encode_kmers and not canonical:
                first = a[: cuts[1] - k + 1]
                mask = (1 << (2 * k)) - 1
                assert bool((((first[:-1] << 2) & mask) >> 2 == (first[1:] >> 2)).all())
            if canonical:
                monkeypatch.delenv("UKM_WIN_STRIP", raising=False)   # the library's own choice (the strip kernel here)
                c = fn(bases, off, k, canonical=canonical)
                assert bool((c == a).all())
                del c
            del a, b


def test_range_fold_equals_chained_fold_large(env, monkeypatch):
    """inter / diff / diff -t over 9 files of ~1.4e7 codes with taxids: the range-partitioned fold kernel (one launch;
    5000+ ranges = several rounds of workgroups, slices of several chunks where a file is denser than the first one)
    against the chained 2-way tile kernels of rounds 1-2 (UKM_NO_FOLD=1) -- two independent device paths, same stream."""
    torch, bench, lib, ctx, A, B = env
    from conftest import synth_tree
    child, parent = synth_tree(6, 8)
    ctx.taxonomy_load(child, parent)
    T = len(child)
    dev = A.device
    nu = 20_000_000
    j = torch.arange(nu, dtype=torch.int64, device=dev)
    U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 9)) & ((1 << 30) - 1)), 0)
    core = (bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 77)) & 7) < 3
    files, taxs = [], []
    for f in range(9):
        h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
        p = 0.55 if f == 0 else (0.95 if f == 4 else 0.7)          # file 4 is much denser than the first: multi-chunk slices
        m = (((h >> 11) & ((1 << 20) - 1)) < int(p * (1 << 20))) | core
        k = U[m]
        files.append(k)
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
    del j, h, m, core
    torch.cuda.synchronize()
    res = {}
    for mode in ("fold", "chain"):
        if mode == "chain":
            monkeypatch.setenv("UKM_NO_FOLD", "1")
        else:
            monkeypatch.delenv("UKM_NO_FOLD", raising=False)
        ik, it = ctx.inter(files, taxs)
        dk, dt = ctx.diff(files, taxs)
        ck, ct = ctx.diff(files, taxs, compare_taxid=True)
        pk = ctx.inter(files)
        res[mode] = [x.clone() for x in (ik, it, dk, dt, ck, ct, pk)]
    monkeypatch.delenv("UKM_NO_FOLD", raising=False)
    assert res["fold"][0].numel() > 1_000_000 and res["fold"][2].numel() > 10 and res["fold"][4].numel() > res["fold"][2].numel()
    for a, b in zip(res["fold"], res["chain"]):
        assert a.numel() == b.numel() and bool((a == b).all())


def test_inter_per_record_taxids_two_launches_equal_the_taxid_kernel_large(env, monkeypatch):
    """Round 5: 2-way `inter` with per-record taxids = the plain-key kernel writing source words + a gather launch
    (UKM_SETOP_SRC=1) against the taxid instantiation (UKM_SETOP_SRC=0, the default since round 6 where one-byte clade codes exist) -- two independent device paths -- on the
    module's two sets (~2e7 records each, 2000+ tiles: source words of every tile shape, pairs split across tile
    boundaries) with uniformly random taxids that include zeros, ids beyond the taxonomy, absent ids inside it and merged
    ids; plain and --mix-taxid; and with a file taxid on one side.  The taxonomy is reloaded in all three clade-code forms
    on the SAME context (one byte + pair table, two bytes + root-path rows of the clade nodes, none)."""
    torch, bench, lib, ctx, A, B = env
    import numpy as np
    dev = A.device

    def tree(kind):
        child, parent = [], []
        if kind == "u8":        # one root, 6-ary, depth 6: 43 nodes of depth <= 2
            nodes = sum(6 ** d for d in range(7))
            child = np.arange(1, nodes + 1, dtype=np.uint32)
            parent = ((child.astype(np.int64) - 2) // 6 + 1).astype(np.uint32)
            parent[0] = 1
        elif kind == "u16":     # 300 roots with a binary tree of depth 5 below each
            per = sum(2 ** d for d in range(6))
            ids = np.arange(1, 300 * per + 1, dtype=np.uint32).reshape(300, per)
            loc = np.arange(per, dtype=np.int64)
            par_loc = np.where(loc == 0, 0, (loc - 1) // 2)
            child = ids.ravel()
            parent = ids[:, par_loc].ravel()
        else:                   # 70,000 roots (no clade table), every 50th with two children
            roots = np.arange(1, 70001, dtype=np.uint32)
            kids = np.arange(70001, 70001 + 2 * 1400, dtype=np.uint32)
            child = np.concatenate([roots, kids])
            parent = np.concatenate([roots, np.repeat(roots[::50], 2)])
        leaf = ~np.isin(child, parent[parent != child])
        keep = np.ones(len(child), dtype=bool)
        keep[np.flatnonzero(leaf)[5::41]] = False   # absent ids inside the range (leaves: nothing hangs below them)
        child, parent = child[keep], parent[keep]
        top = int(child.max())
        mo = np.array([top + 3, top + 4], dtype=np.uint32)
        mn = np.array([child[len(child) // 3], top + 500], dtype=np.uint32)
        return child, parent, mo, mn, top

    for kind in ("u8", "u16", "roots", "u8"):
        child, parent, mo, mn, top = tree(kind)
        ctx.taxonomy_load(child, parent, mo, mn)
        T = top + 6
        ta = (bench.splitmix64_torch(A ^ 12345) & ((1 << 40) - 1)) % T
        tb = (bench.splitmix64_torch(B ^ 54321) & ((1 << 40) - 1)) % T
        ta, tb = ta.to(torch.int32), tb.to(torch.int32)
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("UKM_SETOP_SRC", mode)
            k1, t1 = ctx.setop2(lib.OP_INTER, A, B, ta, tb)
            k2, t2 = ctx.setop2(lib.OP_INTER, A, B, ta, tb, flags=lib.F_MIX_TAXID)
            k3, t3 = ctx.setop2(lib.OP_INTER, A, B, ta, 7)
            res[mode] = [x.clone() for x in (k1, t1, k2, t2, k3, t3)]
        monkeypatch.delenv("UKM_SETOP_SRC", raising=False)
        assert res["1"][0].numel() > 1_000_000
        for x, y in zip(res["1"], res["0"]):
            assert x.numel() == y.numel() and bool((x == y).all()), kind
        # and a slice of it against the oracle-free definition: lca of the two records' taxids through ukm_lca
        k, t = res["1"][0], res["1"][1]
        ia = torch.searchsorted(A, k[:200000])
        ib = torch.searchsorted(B, k[:200000])
        exp = ctx.lca(ta[ia].cpu().numpy().view(np.uint32), tb[ib].cpu().numpy().view(np.uint32))
        assert np.array_equal(t[:200000].cpu().numpy().view(np.uint32), exp), kind


def test_sort_top16_then_lds_buckets(env, monkeypatch):
    """Keys-only sorts of >= 2^24 keys take two scatter passes over the top 16 bits and then sort every bucket inside LDS
    (ukm_sort.hip).  Against torch.sort and against the all-passes route (UKM_SORT_LOCAL=0): evenly spread 62-bit and
    64-bit keys (all buckets small), keys with a few heavy buckets (sorted by the general route afterwards), keys
    crowded into a handful of buckets (the whole call falls back), narrow keys handed over as 64-bit (the OR word narrows
    them), duplicates, all-ones keys, every key the same."""
    torch, bench, lib, ctx, A, B = env
    dev = A.device
    n = 20_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(11)

    def rnd(bits, count=n):
        hi = torch.randint(0, 1 << 31, (count,), device=dev, generator=g, dtype=torch.int64)
        lo = torch.randint(0, 1 << 31, (count,), device=dev, generator=g, dtype=torch.int64)
        x = (hi << 33) ^ (lo << 2) ^ torch.randint(0, 4, (count,), device=dev, generator=g, dtype=torch.int64)
        return x if bits == 64 else x & ((1 << bits) - 1)

    def usort(x):   # unsigned order of the int64 bit patterns
        s = torch.sort(x ^ (-1 << 63)).values
        return s ^ (-1 << 63)

    cases = []
    cases.append((rnd(62), 62))
    cases.append((rnd(64), 64))
    x = rnd(62)
    x[: 3_000_000] = (x[: 3_000_000] & ((1 << 46) - 1)) | (5 << 46)          # one heavy bucket (3e6 keys)
    x[3_000_000: 3_400_000] = (x[3_000_000: 3_400_000] & ((1 << 46) - 1)) | (77 << 46)
    cases.append((x, 62))
    cases.append((rnd(62) & ((1 << 50) - 1) | (3 << 50), 62))                 # 16 buckets only: the general route
    cases.append((rnd(40), 64))                                               # narrow keys declared 64-bit
    y = rnd(62)
    y[::7] = y[1::7][: y[::7].numel()]                                        # duplicates
    y[:5] = -1                                                                # all ones (as 64-bit patterns)
    cases.append((y, 64))
    cases.append((torch.full((n,), 123456789, dtype=torch.int64, device=dev), 62))
    # the buckets' counting step (round 5): keys of a bucket that differ in their lowest 20 bits only share ONE bin (the
    # bucket takes the digit passes), and keys with ~40 copies each fill bins up to and beyond the walk's limit
    cases.append((rnd(62) & ~(((1 << 26) - 1) << 20), 62))
    z = rnd(62, n // 40).repeat(40)
    cases.append((z[torch.randperm(z.numel(), device=dev, generator=g)], 62))
    for x, bits in cases:
        exp = usort(x)
        for knob in (None, "0"):
            if knob is None:
                monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)
            else:
                monkeypatch.setenv("UKM_SORT_LOCAL", knob)
            w = x.clone()
            ctx.sort_u64(w, bits)
            assert torch.equal(w, exp), (bits, knob)
        # taxids riding along: the order of equal keys is the input order (stable)
        order = torch.sort(x ^ (-1 << 63), stable=True).indices
        w = x.clone()
        v = torch.arange(n, dtype=torch.int32, device=dev)
        ctx.sort_pairs(w, v, bits)
        assert torch.equal(w, exp) and torch.equal(v.long(), order), (bits, "pairs")
        del order, w, v
    monkeypatch.delenv("UKM_SORT_LOCAL", raising=False)


def test_sort_three_top_passes_then_lds_buckets(env):
    """Beyond 1.34e8 keys the bucket route sorts by the top 24 bits (three scatter passes, the result sits in the scratch
    copy) and the bucket kernels write the caller's array from there: 1.6e8 random 62-bit keys and 64-bit keys with
    duplicates against torch.sort, keys only and with taxids (stable)."""
    torch, bench, lib, ctx, A, B = env
    dev = A.device
    n = 160_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    for bits in (62, 64):
        hi = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
        lo = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
        x = (hi << 33) ^ (lo << 1)
        del hi, lo
        if bits == 62:
            x &= (1 << 62) - 1
        else:
            x[::5] = x[1::5][: x[::5].numel()]   # duplicates
            x[:3] = -1
        srt = torch.sort(x ^ (-1 << 63), stable=True)
        exp, order = srt.values ^ (-1 << 63), srt.indices
        del srt
        w = x.clone()
        ctx.sort_u64(w, bits)
        assert torch.equal(w, exp), bits
        v = torch.arange(n, dtype=torch.int32, device=dev)
        w.copy_(x)
        ctx.sort_pairs(w, v, bits)
        assert torch.equal(w, exp) and torch.equal(v.long(), order), (bits, "pairs")
        del w, v, exp, order, x
