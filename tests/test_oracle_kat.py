"""Pins the CPU oracle against the reference's own published known answers
(sources: /root/reference/README.md:174-204,270-278,
analysis/distance/README.md:5-9, testdata/old/Ecoli-MG1655.fasta.gz.cr.tsv)."""
import numpy as np
import pytest

from conftest import AMUC, IAI39, MG1655
from oracle import oracle as O


def _distinct_canonical(genomes, name, k):
    bases, off = genomes(name)
    codes = O.count_windows(bases, off, k, hashed=False, canonical=True)
    return O.unique(O.sort_u64(codes), mode=O.UNIQUE)


@pytest.fixture(scope="module")
def k23(genomes):
    return {n: _distinct_canonical(genomes, n, 23) for n in (MG1655, IAI39, AMUC)}


def test_c1_distinct_canonical_23mers(k23):
    # README.md:200-204
    assert len(k23[MG1655]) == 4546632
    assert len(k23[IAI39]) == 4902266
    assert len(k23[AMUC]) == 2630905


def test_c2_first_sorted_kmers(k23):
    # README.md:177-180
    first = [O.decode(int(c), 23) for c in k23[MG1655][:3]]
    assert first == ["AAAAAAAAACCATCCAAATCTGG", "AAAAAAAAACCGCTAGTATATTC", "AAAAAAAAACCTGAAAAAAACGG"]
    assert [int(c) for c in k23[MG1655][:3]] == [87360378, 94155581, 98566170]
    assert O.encode("AAAAAAAAACCATCCAAATCTGG") == 87360378


def test_c3_set_cardinalities(k23):
    # README.md:270-278
    a, b = k23[IAI39], k23[MG1655]
    assert len(O.union([a, b])) == 6872728
    assert len(O.inter([a, b])) == 2576170
    assert len(O.diff([a, b])) == 2326096
    # `sort -d` of the concatenation (dup.k23.unik)
    cat = O.sort_u64(np.concatenate([a, b]))
    assert len(O.unique(cat, mode=O.REPEATED)) == 2576170
    # common with threshold 2 == inter for two sets
    assert np.array_equal(O.common([a, b], 2), O.inter([a, b]))
    # numpy cross-check of the actual streams
    assert np.array_equal(O.union([a, b]), np.union1d(a, b))
    assert np.array_equal(O.inter([a, b]), np.intersect1d(a, b))
    assert np.array_equal(O.diff([a, b]), np.setdiff1d(a, b))


@pytest.mark.parametrize("k,expected", [(21, 4543891), (31, 4554269)])
def test_c4_distinct_counts_other_k(genomes, k, expected):
    # testdata/old/Ecoli-MG1655.fasta.gz.cr.tsv rows K=21, 31 (plain bytes / (k+1))
    assert len(_distinct_canonical(genomes, MG1655, k)) == expected


def test_c5_nthash_values():
    # README.md:183-186
    kat = {
        "CATCCGCCATCTTTGGGGTGTCG": (12969044065694723203, 1210726578792),
        "AGCGCAAAATCCCCAAACATGTA": (2286899379883, 14136929502914076711),
        "AACTGATTTTTGATGATGACTCC": (3542156397282, 7314948916677284658),
    }
    for kmer, (fwd, rev) in kat.items():
        assert O.nthash_kmer(kmer) == (fwd, rev)
        assert int(O.hash_iter(kmer, 23, canonical=True)[0]) == min(fwd, rev)
        assert int(O.hash_iter(kmer, 23, canonical=False)[0]) == fwd


def test_c10_max_hash_constants():
    # count.go:98
    assert O.max_hash(1000) == 18446744073709552
    assert O.max_hash(15) == 1229782938247303424


def test_c6_scaled_minhash_count(genomes):
    # analysis/distance/README.md:9,47-48 : count -k 31 -K -s -H -D 15
    bases, off = genomes(MG1655)
    h = O.count_windows(bases, off, 31, hashed=True, canonical=True)
    distinct = O.unique(O.sort_u64(h), mode=O.UNIQUE)
    assert len(distinct) == 4554269
    kept = O.count_windows(bases, off, 31, hashed=True, canonical=True, max_hash=O.max_hash(15))
    assert len(O.unique(O.sort_u64(kept), mode=O.UNIQUE)) == 586734
    assert np.array_equal(O.sort_u64(kept), O.sort_u64(h[h <= np.uint64(O.max_hash(15))]))


def test_c7_minimizer_linear(genomes):
    # README.md:174,183-194,199 : count -k 23 -W 5 -H -K -l
    bases, off = genomes(AMUC)
    h, pos = O.minimizer(bases, 23, 5)
    assert len(h) == 860900
    assert [int(p) for p in pos[:5]] == [2, 5, 6, 9, 13]
    # README.md:183-186 lists the three smallest hashes of that file (a sorted view)
    assert [int(x) for x in np.sort(h)[:3]] == [1210726578792, 2286899379883, 3542156397282]
    assert bytes(bases[2:25]).decode() == "ATCTTATAAAATAACCACATAAC"


def test_c8_minimizer_distinct(genomes):
    # analysis/distance/README.md:8,41-42 : count -k 31 -K -s -H -W 15
    bases, off = genomes(MG1655)
    h, _ = O.minimizer(bases, 31, 15)
    assert len(O.unique(O.sort_u64(h), mode=O.UNIQUE)) == 549963


def test_rolling_equals_direct(genomes):
    bases, _ = genomes(AMUC)
    seq = bases[:5000]
    for k in (1, 5, 23, 31, 32):
        it = O.kmer_iter(seq, k, canonical=True)
        for i in (0, 1, 17, len(it) - 1):
            c = O.encode(bytes(seq[i:i + k]))
            assert int(it[i]) == O.canonical(c, k)
            assert O.revcomp(O.revcomp(c, k), k) == c
    for k in (1, 23, 51, 63, 64):
        it = O.hash_iter(seq, k, canonical=True)
        itf = O.hash_iter(seq, k, canonical=False)
        for i in (0, 1, 99, len(it) - 1):
            f, r = O.nthash_kmer(bytes(seq[i:i + k]))
            assert int(it[i]) == min(f, r)
            assert int(itf[i]) == f


def test_circular_and_short(genomes):
    bases, _ = genomes(AMUC)
    seq = bases[:300]
    k = 21
    circ = O.kmer_iter(seq, k, canonical=False, circular=True)
    lin = O.kmer_iter(np.concatenate([seq, seq[:k - 1]]), k, canonical=False)
    assert len(circ) == len(seq) and np.array_equal(circ, lin)
    with pytest.raises(ValueError):
        O.kmer_iter(seq[:10], k)
    with pytest.raises(ValueError):
        O.kmer_iter(b"ACGTXACGT", 3)
    # degenerate bases collapse to their first base (kmers v0.1.0): N->A, Y->C, K->G
    assert O.encode("NYK") == O.encode("ACG")
