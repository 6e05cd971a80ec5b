"""Tests of the `unikmer`-compatible C++ driver (unikmer_amd/bin/unikmer).

CPU part: the commands that never touch the GPU (dump / view / num / info / concat / head /
encode / decode) — `.unik` container round trips in every body encoding.
GPU part (-m gpu): the README quick-start transcript (README.md:154-278) replayed through the
binary on the three fixture genomes: count -> sort/union/inter/diff/common/merge -> num/view.
"""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import AMUC, GOLDEN, IAI39, MG1655

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "unikmer_amd", "bin", "unikmer")


@pytest.fixture(scope="module")
def cli():
    from unikmer_amd import build
    build.build()
    assert os.path.exists(BIN)

    def run(*args, stdin=None, ok=True):
        p = subprocess.run([BIN] + [str(a) for a in args], input=stdin, capture_output=True)
        if ok:
            assert p.returncode == 0, p.stderr.decode()
        return p
    return run


def _kmers(n, k, seed=0):
    rng = np.random.default_rng(seed)
    codes = np.unique(rng.integers(0, 4 ** k, n, dtype=np.uint64))
    def dec(c):
        return "".join("ACGT"[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k))
    return codes, [dec(c) for c in codes]


@pytest.mark.parametrize("flags", [[], ["-s"], ["-c"], ["-C"], ["-s", "-C"]])
def test_dump_view_roundtrip(cli, tmp_path, flags):
    codes, kmers = _kmers(5001, 21)   # odd count exercises the trailing single record of sorted files
    txt = ("\n".join(kmers) + "\n").encode()
    out = tmp_path / "a"
    cli("dump", *flags, "-o", out, stdin=txt)
    f = str(out) + ".unik"
    assert cli("view", f).stdout == txt
    assert cli("view", "-N", f).stdout.decode().split() == [str(int(c)) for c in codes]
    assert cli("num", "-f", f).stdout.strip() == b"5001"
    raw = open(f, "rb").read()
    assert (raw[:2] == b"\x1f\x8b") == ("-C" not in flags)      # gzip unless -C (util-io.go:57-64)
    info = cli("info", "-a", "--symbol-true", "T", "--symbol-false", "F", f).stdout.decode().splitlines()[1].split("\t")
    assert info[1] == "21" and info[7] == ("T" if "-s" in flags else "F") and info[8] == ("T" if "-c" in flags else "F")


def test_dump_canonical_taxid_unique(cli, tmp_path):
    lines = ["ACGTACGTAC\t5", "GTACGTACGT\t7", "AAAAAAAAAA\t9", "ACGTACGTAC\t5"]   # line 2 = revcomp of line 1
    out = tmp_path / "t"
    cli("dump", "-K", "-u", "-o", out, stdin=("\n".join(lines) + "\n").encode())
    got = cli("view", "-t", str(out) + ".unik").stdout.decode().splitlines()
    assert got == ["ACGTACGTAC\t5", "AAAAAAAAAA\t9"]
    assert cli("view", "-T", str(out) + ".unik").stdout.split() == [b"5", b"9"]
    # -O keeps only k-mers that are already canonical
    cli("dump", "-O", "-o", out, stdin=b"GTACGTACGT\nACGTACGTAC\n")
    assert cli("view", str(out) + ".unik").stdout == b"ACGTACGTAC\n"
    # illegal base -> error exit like checkError (util-cli.go:39-44)
    p = cli("dump", "-o", out, stdin=b"ACGTXCGTAC\n", ok=False)
    assert p.returncode == 255 and b"fail to encode" in p.stderr


def test_sorted_taxid_body_and_global_taxid(cli, tmp_path):
    codes, kmers = _kmers(2000, 15, seed=3)
    txt = "".join("%s\t%d\n" % (k, 1000 + i % 70000) for i, k in enumerate(kmers)).encode()
    out = tmp_path / "st"
    cli("dump", "-s", "--max-taxid", 100000, "-o", out, stdin=txt)
    assert cli("view", "-t", str(out) + ".unik").stdout == txt
    out2 = tmp_path / "g"
    cli("dump", "-t", 511145, "-o", out2, stdin=("\n".join(kmers) + "\n").encode())
    lines = cli("view", "-t", str(out2) + ".unik").stdout.decode().splitlines()
    assert lines[0].endswith("\t511145") and len(lines) == len(kmers)
    assert "511145" in cli("info", str(out2) + ".unik").stdout.decode()


def test_concat_head_encode_decode(cli, tmp_path):
    c1, k1 = _kmers(300, 11, seed=1)
    c2, k2 = _kmers(200, 11, seed=2)
    a, b = tmp_path / "a", tmp_path / "b"
    cli("dump", "-o", a, stdin=("\n".join(k1) + "\n").encode())
    cli("dump", "-o", b, stdin=("\n".join(k2) + "\n").encode())
    cli("concat", "-o", tmp_path / "c", str(a) + ".unik", str(b) + ".unik")
    assert cli("view", str(tmp_path / "c") + ".unik").stdout.decode().split() == k1 + k2
    assert cli("num", str(tmp_path / "c") + ".unik").stdout.strip() == b"-1"     # README.md:269
    cli("head", "-n", 7, "-o", tmp_path / "h", str(a) + ".unik")
    assert cli("view", str(tmp_path / "h") + ".unik").stdout.decode().split() == k1[:7]
    enc = cli("encode", stdin=b"AAAAAAAAACCATCCAAATCTGG\n").stdout.strip()
    assert enc == b"87360378"                                                    # README.md:177-180 / SURVEY C-2
    assert cli("decode", "-k", 23, stdin=enc + b"\n").stdout.strip() == b"AAAAAAAAACCATCCAAATCTGG"
    assert cli("encode", "-K", stdin=b"TTTT\n").stdout.strip() == b"0"
    # k mismatch between files -> error (util-binary-file.go:31-44)
    cli("dump", "-o", tmp_path / "k9", stdin=b"ACGTACGTA\n")
    p = cli("concat", "-o", tmp_path / "x", str(a) + ".unik", str(tmp_path / "k9") + ".unik", ok=False)
    assert p.returncode == 255 and b"k-mer length not consistent" in p.stderr


def test_unknown_command_and_flags(cli):
    assert cli("frobnicate", ok=False).returncode == 255
    assert cli("view", "--no-such-flag", ok=False).returncode == 255
    assert b"unikmer v" in cli("version").stdout


# --------------------------------------------------------------------------------------------- GPU
def _fa(name):
    return os.path.join(GOLDEN, name)


@pytest.mark.gpu
def test_readme_transcript_on_gpu(cli, tmp_path):
    d = str(tmp_path)
    mg, ia, am = d + "/mg", d + "/ia", d + "/am"
    # count -k 23 -K -s with global taxids (README.md:167-171)
    cli("count", "-k", 23, "-K", "-s", _fa(MG1655), "-o", mg, "-t", 511145)
    cli("count", "-k", 23, "-K", "-s", _fa(IAI39), "-o", ia, "-t", 585057)
    cli("count", "-k", 23, "-K", "-s", _fa(AMUC), "-o", am, "-t", 349741)
    nums = cli("num", mg + ".unik", ia + ".unik", am + ".unik").stdout.split()
    assert nums == [b"4546632", b"4902266", b"2630905"]                          # README.md:200-204
    head = cli("view", "--show-taxid", mg + ".unik").stdout.decode().splitlines()[:3]
    assert head == ["AAAAAAAAACCATCCAAATCTGG\t511145", "AAAAAAAAACCGCTAGTATATTC\t511145",
                    "AAAAAAAAACCTGAAAAAAACGG\t511145"]                           # README.md:177-180
    # minimizers in linear order (README.md:173-174,199: 860,900) and their three smallest hashes (README.md:183-186)
    cli("count", "-k", 23, "-W", 5, "-H", "-K", "-l", _fa(AMUC), "-o", d + "/am.m")
    assert cli("num", d + "/am.m.unik").stdout.strip() == b"860900"
    mins = sorted(int(x) for x in cli("view", d + "/am.m.unik").stdout.split())
    assert mins[:3] == [1210726578792, 2286899379883, 3542156397282]
    # unsorted + compact count gives the same set (README.md:154-158)
    cli("count", "-k", 23, _fa(MG1655), "-o", d + "/mgc", "--canonical", "--compact")
    assert cli("num", d + "/mgc.unik").stdout.strip() == b"4546632"
    # set operations without taxonomy (-I ignores the global taxids; LCA needs NCBI nodes.dmp)
    cli("union", "-I", "-s", ia + ".unik", mg + ".unik", "-o", d + "/union")
    cli("inter", "-I", ia + ".unik", mg + ".unik", "-o", d + "/inter")
    cli("diff", "-I", "-s", ia + ".unik", mg + ".unik", "-o", d + "/diff")
    cli("common", "-I", ia + ".unik", mg + ".unik", "-o", d + "/common")
    cli("concat", "-I", ia + ".unik", mg + ".unik", "-o", d + "/concat")
    cli("sort", "-I", "-u", d + "/concat.unik", "-o", d + "/union2")
    cli("sort", "-I", "-d", d + "/concat.unik", "-o", d + "/dup")
    cli("merge", "-I", "-u", ia + ".unik", mg + ".unik", "-o", d + "/union3")
    n = {k: int(cli("num", "-f", d + "/%s.unik" % k).stdout) for k in ("union", "inter", "diff", "common", "union2", "dup", "union3")}
    assert n == {"union": 6872728, "inter": 2576170, "diff": 2326096, "common": 2576170,
                 "union2": 6872728, "dup": 2576170, "union3": 6872728}            # README.md:270-278
    # README.md:215-229: union -s == sort -u (same md5 of the text view)
    md5 = {k: hashlib.md5(cli("view", d + "/%s.unik" % k).stdout).hexdigest() for k in ("union", "union2", "union3")}
    assert md5["union"] == md5["union2"] == md5["union3"]
    assert hashlib.md5(cli("view", d + "/inter.unik").stdout).hexdigest() == hashlib.md5(cli("view", d + "/dup.unik").stdout).hexdigest()


@pytest.mark.gpu
def test_count_modes_on_gpu(cli, tmp_path):
    d = str(tmp_path)
    seq = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTAGCTAGGATCCGATCGTTAGC"
    with gzip.open(d + "/x.fa.gz", "wt") as fh:
        fh.write(">r1 taxid=9\n%s\n>r2 taxid=9\n%s\n>short\nACG\n" % (seq, seq[:30]))
    k = 11
    # -l linear: every window of every record with len >= k, in order (count.go:413-422)
    cli("count", "-k", k, "-l", d + "/x.fa.gz", "-o", d + "/lin")
    lin = cli("view", d + "/lin.unik").stdout.decode().split()
    exp = [seq[i:i + k] for i in range(len(seq) - k + 1)] + [seq[i:i + k] for i in range(30 - k + 1)]
    assert lin == exp
    # -d / -u partition the distinct set (count.go:424-432)
    cli("count", "-k", k, "-s", d + "/x.fa.gz", "-o", d + "/all")
    cli("count", "-k", k, "-s", "-d", d + "/x.fa.gz", "-o", d + "/rep")
    cli("count", "-k", k, "-s", "-u", d + "/x.fa.gz", "-o", d + "/uni")
    allk = set(cli("view", d + "/all.unik").stdout.decode().split())
    rep = set(cli("view", d + "/rep.unik").stdout.decode().split())
    uni = set(cli("view", d + "/uni.unik").stdout.decode().split())
    assert allk == set(exp) and rep == set(exp[len(seq) - k + 1:]) and rep | uni == allk and not (rep & uni)
    # hashed + scaled (count.go:94-98,373-375): header carries the scale, values <= maxHash
    cli("count", "-k", k, "-K", "-s", "-H", "-D", 3, d + "/x.fa.gz", "-o", d + "/sc")
    info = cli("info", "--symbol-true", "T", "--symbol-false", "F", d + "/sc.unik").stdout.decode().splitlines()[1].split("\t")
    assert info[3] == "T" and info[4] == "T"
    vals = [int(x) for x in cli("view", d + "/sc.unik").stdout.split()]
    assert vals == sorted(vals) and all(v <= (2 ** 64 - 1) // 3 + 1 for v in vals) and len(vals) > 0
    # FASTQ input
    with open(d + "/y.fq", "w") as fh:
        fh.write("@q1\n%s\n+\n%s\n" % (seq, "I" * len(seq)))
    cli("count", "-k", k, "-l", d + "/y.fq", "-o", d + "/fq")
    assert cli("view", d + "/fq.unik").stdout.decode().split() == exp[:len(seq) - k + 1]


@pytest.mark.gpu
def test_count_device_pipeline_chunked(cli, tmp_path, monkeypatch):
    """`count` keeps the window values on the device (chunked, double-buffered upload; sort + unique there):
    with 1 MB chunks a 6 Mbp multi-record FASTA goes through several chunks, incl. a record longer than a chunk,
    an empty record and records shorter than k.  Result = the oracle's count on the same records."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    lens = [1_500_000, 0, 20, 700_000, 31, 30, 2_000_000] + [150] * 4000 + [1_200_000]
    seqs = ["".join(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes().decode()) for n in lens]
    d = str(tmp_path)
    with open(d + "/m.fa", "w") as fh:
        for i, q in enumerate(seqs):
            fh.write(">r%d\n" % i)
            for j in range(0, len(q), 70):
                fh.write(q[j:j + 70] + "\n")
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    monkeypatch.setenv("UNIKMER_CHUNK_MB", "1")
    k = 31
    p = cli("count", "-k", k, "-K", "-s", "--verbose", d + "/m.fa", "-o", d + "/c")
    assert b"device pipeline:" in p.stderr and b" 1 chunk(s)" not in p.stderr
    got = np.array([int(x) for x in cli("view", "-N", d + "/c.unik").stdout.split()], dtype=np.uint64)
    exp = O.unique(O.sort_u64(O.count_windows(bases, off, k, canonical=True)))
    assert np.array_equal(got, exp)
    # linear output keeps window order across chunk boundaries; hashed + scaled goes through the same pipeline
    cli("count", "-k", k, "-l", d + "/m.fa", "-o", d + "/l")
    got = np.array([int(x) for x in cli("view", "-N", d + "/l.unik").stdout.split()], dtype=np.uint64)
    assert np.array_equal(got, O.count_windows(bases, off, k, canonical=False))
    cli("count", "-k", 51, "-K", "-H", "-s", "-D", 200, d + "/m.fa", "-o", d + "/h")
    got = np.array([int(x) for x in cli("view", "-N", d + "/h.unik").stdout.split()], dtype=np.uint64)
    exp = O.unique(O.sort_u64(O.count_windows(bases, off, 51, hashed=True, canonical=True, max_hash=O.max_hash(200))))
    assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_count_taxid_and_minimizer_stream_through_chunks(cli, tmp_path, monkeypatch):
    """`count -T -r` (a taxid per record, parsed from the header: count.go:334-344) and `count -W` go through the same
    chunked device pipeline as the plain path since round 3: 1 MB chunks, records shorter than k (skipped BEFORE their
    header is parsed: the header without a taxid must not fail), a record longer than a chunk, `-B` name filter.
    Expected: the oracle's windows with each record's taxid, sorted, LCA-folded per code; the oracle's minimizers."""
    from conftest import synth_tree
    from oracle import oracle as O
    d = str(tmp_path)
    child, parent = synth_tree(depth=4, arity=4)
    os.makedirs(d + "/tax")
    with open(d + "/tax/nodes.dmp", "w") as fh:
        for c, p in zip(child, parent):
            fh.write("%d\t|\t%d\t|\tno rank\t|\n" % (c, p))
    tax = O.Taxonomy(child, parent)
    T = len(child)
    rng = np.random.default_rng(9)
    k = 21
    lens = [1_300_000, 20, 5000, 0, 21, 400_000] + [150] * 3000 + [900_000]
    # a small alphabet window makes codes repeat across records, so the LCA fold has work
    pool = "".join(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200_000)].tobytes().decode())
    seqs, taxid = [], []
    for n in lens:
        st = int(rng.integers(0, max(1, len(pool) - 1)))
        q = (pool[st:] + pool * (n // len(pool) + 1))[:n]
        seqs.append(q)
        taxid.append(int(rng.integers(1, T + 1)))
    with open(d + "/t.fa", "w") as fh:
        for i, q in enumerate(seqs):
            short = len(q) < k
            fh.write((">r%d notaxid\n" % i) if short else (">r%d taxid|%d| skipme%d\n" % (i, taxid[i], i % 7)))
            for j in range(0, len(q), 80):
                fh.write(q[j:j + 80] + "\n")
    monkeypatch.setenv("UNIKMER_CHUNK_MB", "1")
    keep = [i for i in range(len(seqs)) if i % 7 != 3 or len(seqs[i]) < k]      # -B drops "skipme3"
    bases = np.frombuffer("".join(seqs[i] for i in keep).encode(), dtype=np.uint8)
    off = np.zeros(len(keep) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(seqs[i]) for i in keep])
    p = cli("count", "-k", k, "-K", "-s", "-T", "-r", r"taxid\|(\d+)", "-B", "skipme3", "--data-dir", d + "/tax", "--verbose",
            d + "/t.fa", "-o", d + "/ct")
    assert b"device pipeline:" in p.stderr and b" 1 chunk(s)" not in p.stderr
    gk = np.array([int(x) for x in cli("view", "-N", d + "/ct.unik").stdout.split()], dtype=np.uint64)
    gt = np.array([int(x) for x in cli("view", "-T", d + "/ct.unik").stdout.split()], dtype=np.uint32)
    codes = O.count_windows(bases, off, k, canonical=True)
    wt = np.concatenate([np.full(max(0, len(seqs[i]) - k + 1), taxid[i], dtype=np.uint32) for i in keep]) if keep else np.empty(0, np.uint32)
    assert len(codes) == len(wt)
    sk, st_ = O.sort_pairs(codes, wt)
    ek, et = O.unique(sk, st_, tax=tax)
    assert np.array_equal(gk, ek) and np.array_equal(gt, et)
    # a header without a taxid on a record that HAS windows is an error (count.go:338-341)
    with open(d + "/bad.fa", "w") as fh:
        fh.write(">x notaxid\n" + seqs[2] + "\n")
    bad = cli("count", "-k", k, "-K", "-s", "-T", "-r", r"taxid\|(\d+)", "--data-dir", d + "/tax", d + "/bad.fa", "-o", d + "/b", ok=False)
    assert bad.returncode != 0 and b"failed to parse taxid" in bad.stderr
    # -W: the minimizer sketch record by record through the chunks (linear order), then as a sorted set
    cli("count", "-k", k, "-W", 11, "-H", "-K", "-l", d + "/t.fa", "-o", d + "/mw")
    got = np.array([int(x) for x in cli("view", "-N", d + "/mw.unik").stdout.split()], dtype=np.uint64)
    all_b = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    all_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    all_off[1:] = np.cumsum(lens)
    exp = np.concatenate([O.minimizer(all_b[int(all_off[i]):int(all_off[i + 1])], k, 11)[0] for i in range(len(seqs))
                          if lens[i] >= k + 11 - 1])   # shorter records have no group of w windows: ErrShortSeq, skipped
    assert np.array_equal(got, exp)
    cli("count", "-k", k, "-W", 11, "-H", "-K", "-s", d + "/t.fa", "-o", d + "/ms")
    got = np.array([int(x) for x in cli("view", "-N", d + "/ms.unik").stdout.split()], dtype=np.uint64)
    assert np.array_equal(got, np.unique(exp))


@pytest.mark.gpu
def test_taxonomy_lca_through_cli(cli, tmp_path):
    """union / inter / sort -u / diff -t with per-record taxids and a synthetic nodes.dmp +
    merged.dmp (util.go:119-171); expected values from the oracle."""
    from conftest import synth_tree
    from oracle import oracle as O
    d = str(tmp_path)
    child, parent = synth_tree(depth=4, arity=4)
    os.makedirs(d + "/tax")
    with open(d + "/tax/nodes.dmp", "w") as fh:
        for c, p in zip(child, parent):
            fh.write("%d\t|\t%d\t|\tno rank\t|\n" % (c, p))
    with open(d + "/tax/merged.dmp", "w") as fh:
        fh.write("9000\t|\t7\t|\n")
    tax = O.Taxonomy(child, parent, [9000], [7])
    T = len(child)
    rng = np.random.default_rng(5)
    k = 13
    files, taxs = [], []
    for f in range(3):
        codes = np.unique(rng.integers(0, 3000, 1500).astype(np.uint64))
        t = rng.integers(1, T + 1, len(codes)).astype(np.uint32)
        t[::50] = 9000                                   # a merged id
        files.append(codes)
        taxs.append(t)
        kmers = ["".join("ACGT"[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k)) for c in codes]
        txt = "".join("%s\t%d\n" % (km, tt) for km, tt in zip(kmers, t)).encode()
        cli("dump", "-s", "-o", d + "/f%d" % f, stdin=txt)
    fs = [d + "/f%d.unik" % f for f in range(3)]

    def view(name):
        out = cli("view", "-N", name).stdout.split()
        tx = cli("view", "-T", name).stdout.split()
        return np.array([int(x) for x in out], dtype=np.uint64), np.array([int(x) for x in tx], dtype=np.uint32)

    cli("union", "-s", "--data-dir", d + "/tax", *fs, "-o", d + "/u")
    gk, gt = view(d + "/u.unik")
    ok, ot = O.union(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    cli("inter", "--data-dir", d + "/tax", *fs, "-o", d + "/i")
    gk, gt = view(d + "/i.unik")
    ok, ot = O.inter(files, taxs, tax)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    cli("diff", "-s", "-t", "--data-dir", d + "/tax", *fs, "-o", d + "/d")
    gk, gt = view(d + "/d.unik")
    ok, ot = O.diff(files, taxs, tax, compare_taxid=True)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    cli("concat", *fs, "-o", d + "/c")
    cli("sort", "-u", "--data-dir", d + "/tax", d + "/c.unik", "-o", d + "/su")
    gk, gt = view(d + "/su.unik")
    ok, ot = O.union(files, taxs, tax)                   # README.md:215-229 equivalence
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    # the same through the chunk protocol (per-chunk LCA fold, then LCA fold in both merge rounds)
    cli("sort", "-u", "-m", 500, "-M", 3, "-t", d, "--data-dir", d + "/tax", d + "/c.unik", "-o", d + "/su2")
    gk, gt = view(d + "/su2.unik")
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
    cli("sort", "-d", "--data-dir", d + "/tax", d + "/c.unik", "-o", d + "/sd")
    cli("sort", "-d", "-m", 500, "-M", 3, "-t", d, "--data-dir", d + "/tax", d + "/c.unik", "-o", d + "/sd2")
    k1, t1 = view(d + "/sd.unik")
    k2, t2 = view(d + "/sd2.unik")
    assert len(k1) > 0 and np.array_equal(k1, k2) and np.array_equal(t1, t2)
    # missing taxonomy directory -> error, not a silent zero
    p = cli("union", "-s", "--data-dir", d + "/nope", *fs, "-o", d + "/x", ok=False)
    assert p.returncode == 255 and b"taxonomy file not found" in p.stderr


@pytest.mark.gpu
def test_sort_chunk_protocol_split_merge_on_gpu(cli, tmp_path):
    """SURVEY C-9 / README.md:215-229: `sort -u` == `sort -u -m N` == `split -u -m N` + `merge -u -D`, and the
    same for -d (two-copy chunk protocol, util-sort.go:70-91,377-388) and for the plain sort; small
    --max-open-files forces the two merge rounds of sort.go:365-420."""
    d = str(tmp_path)
    rng = np.random.default_rng(11)
    k = 15
    a = rng.integers(0, 3000, 5000)     # many duplicates, also across chunk borders
    b = rng.integers(1000, 6000, 4000)
    for name, arr in (("a", a), ("b", b)):
        txt = "\n".join("".join("ACGT"[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k)) for c in arr) + "\n"
        cli("dump", "-o", d + "/" + name, stdin=txt.encode())
    ins = [d + "/a.unik", d + "/b.unik"]
    view = lambda f: cli("view", f).stdout
    for flag in ("-u", "-d", None):
        fl = [flag] if flag else []
        tag = (flag or "p").strip("-")
        cli("sort", *fl, *ins, "-o", d + "/whole_" + tag)
        ref = view(d + "/whole_%s.unik" % tag)
        assert len(ref) > 0
        # chunked: 700 k-mers per chunk -> 13 chunks; 2 rounds when --max-open-files 4
        for mo in (400, 4):
            cli("sort", *fl, "-m", 700, "-M", mo, "-t", d, *ins, "-o", d + "/chunk_%s_%d" % (tag, mo))
            assert view(d + "/chunk_%s_%d.unik" % (tag, mo)) == ref
            assert not os.path.exists(d + "/chunk_%s_%d.tmp" % (tag, mo))          # tmp dir removed
        # split + merge
        cli("split", *fl, "-m", "1K", "-O", d + "/split_" + tag, *ins)
        chunks = sorted(os.listdir(d + "/split_" + tag))
        assert chunks[0] == "chunk_001.unik" and len(chunks) == 9                  # 9000 k-mers / 1024
        cli("merge", *fl, "-D", d + "/split_" + tag, "-o", d + "/merged_" + tag)
        assert view(d + "/merged_%s.unik" % tag) == ref
        cli("merge", *fl, "-M", 3, "-t", d, *[d + "/split_%s/%s" % (tag, c) for c in chunks], "-o", d + "/merged2_" + tag)
        assert view(d + "/merged2_%s.unik" % tag) == ref
    # expected contents from first principles
    allc = np.concatenate([a, b])
    vals, cnt = np.unique(allc, return_counts=True)
    dec = lambda f: [int(x) for x in cli("view", "--show-code-only", f).stdout.split()]
    assert dec(d + "/whole_u.unik") == [int(v) for v in vals]
    assert dec(d + "/whole_d.unik") == [int(v) for v in vals[cnt > 1]]
    assert dec(d + "/whole_p.unik") == sorted(int(v) for v in allc)
    # a non-empty tmp dir needs --force (sort.go:119-132)
    os.makedirs(d + "/x.tmp"); open(d + "/x.tmp/junk", "w").close()
    p = cli("sort", "-m", 700, "-t", d, *ins, "-o", d + "/x", ok=False)
    assert p.returncode == 255 and b"not empty" in p.stderr
    cli("sort", "-m", 700, "-t", d, "--force", *ins, "-o", d + "/x")
    assert view(d + "/x.unik") == view(d + "/whole_p.unik")


@pytest.mark.gpu
def test_hashed_text_paths_on_gpu(cli, tmp_path):
    """encode -H / dump -H (encode.go:106-113, dump.go:250-275: the ntHash of whole k-mers given as text) and
    view -g (view.go:139-183 + loadHash2Loc util.go:344-393: hashed canonical k-mers decoded back to sequence
    through the genomes, first occurrence in circular records wins) — all computed by the HIP library."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    k = 41
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, k)) for _ in range(300)]
    seqs[7] = seqs[7][:10] + "N" + seqs[7][11:]          # a non-ACGT base hashes with the zero seed
    seqs[9] = seqs[3]                                     # a repeated k-mer
    txt = "\n".join(seqs) + "\n"
    open(d + "/k.txt", "w").write(txt)

    def oracle_hash(s, canonical):
        b = np.frombuffer(s.encode(), dtype=np.uint8)
        return int(O.count_windows(b, np.array([0, len(b)], dtype=np.uint64), len(b), hashed=True, canonical=canonical)[0])
    for canonical in (False, True):
        flags = ["-K"] if canonical else []
        got = [int(x) for x in cli("encode", "-H", *flags, d + "/k.txt").stdout.split()]
        assert got == [oracle_hash(s, canonical) for s in seqs]
    # dump -H -K -u -> hashed canonical .unik; view prints the hashes
    cli("dump", "-H", "-K", "-u", d + "/k.txt", "-o", d + "/h")
    info = cli("info", "-a", d + "/h.unik").stdout.decode()
    assert "41" in info
    hashes = [int(x) for x in cli("view", d + "/h.unik").stdout.split()]
    exp = []
    for s in seqs:
        h = oracle_hash(s, True)
        if h not in exp:
            exp.append(h)
    assert hashes == exp and len(exp) == len(seqs) - 1
    # mismatching lengths are an error (encode.go:101-103)
    open(d + "/bad.txt", "w").write(seqs[0] + "\n" + seqs[1][:-1] + "\n")
    assert cli("encode", "-H", d + "/bad.txt", ok=False).returncode != 0
    # view -g: a minimizer sketch of a genome decoded back to k-mers of that genome
    cli("count", "-k", 31, "-W", 50, "-H", "-K", "-s", _fa(AMUC), "-o", d + "/m")
    codes = [int(x) for x in cli("view", d + "/m.unik").stdout.split()]
    both = cli("view", "-n", "-g", _fa(AMUC), d + "/m.unik").stdout.decode().splitlines()
    assert len(both) == len(codes) > 1000
    for line, c in list(zip(both, codes))[:2000]:
        km, cc = line.split("\t")
        assert int(cc) == c and len(km) == 31 and oracle_hash(km, True) == c
    # a code that is in no genome is printed as the integer (view.go:176-181)
    open(d + "/one.txt", "w").write("12345\n")
    cli("dump", "--hashed", "-k", 31, d + "/one.txt", "-o", d + "/one")
    p = cli("view", "-g", _fa(AMUC), d + "/one.unik")
    assert p.stdout.decode().strip() == "12345" and b"not found in given genomes" in p.stderr


@pytest.mark.gpu
def test_global_taxids_through_set_operations(cli, tmp_path):
    """The reference's documented taxid workflow (README.md:170: `count ... -t 511145`): ONE taxid per file in the .unik
    header (count.go:466-468), handed out with every record by the reader and folded by union / inter / diff -t / common /
    merge like a per-record taxid.  The driver passes it to the library as one number per file (ukm_*_ft); expected values:
    the oracle over the EXPANDED arrays.  One file with per-record taxids beside the global ones, too."""
    from conftest import synth_tree
    from oracle import oracle as O
    d = str(tmp_path)
    child, parent = synth_tree(depth=4, arity=4)
    os.makedirs(d + "/tax")
    with open(d + "/tax/nodes.dmp", "w") as fh:
        for c, p in zip(child, parent):
            fh.write("%d\t|\t%d\t|\tno rank\t|\n" % (c, p))
    with open(d + "/tax/merged.dmp", "w") as fh:
        fh.write("9000\t|\t7\t|\n")
    tax = O.Taxonomy(child, parent, [9000], [7])
    T = len(child)
    rng = np.random.default_rng(11)
    k = 13
    gtax = [int(T - 3), int(T - 2), 9000, 7, int(T - 40)]           # siblings, a merged id and its target, a cousin
    files, taxs = [], []
    for f in range(5):
        codes = np.unique(rng.integers(0, 4000, 2200).astype(np.uint64))
        files.append(codes)
        kmers = ["".join("ACGT"[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k)) for c in codes]
        if f == 4:   # per-record taxids in the last file
            t = rng.integers(1, T + 1, len(codes)).astype(np.uint32)
            txt = "".join("%s\t%d\n" % (km, tt) for km, tt in zip(kmers, t)).encode()
            cli("dump", "-s", "-o", d + "/g%d" % f, stdin=txt)
        else:
            t = np.full(len(codes), gtax[f], np.uint32)
            cli("dump", "-s", "-t", gtax[f], "-o", d + "/g%d" % f, stdin=("\n".join(kmers) + "\n").encode())
        taxs.append(t)
    fs = [d + "/g%d.unik" % f for f in range(5)]

    def view(name):
        out = cli("view", "-N", name).stdout.split()
        tx = cli("view", "-T", name).stdout.split()
        return np.array([int(x) for x in out], dtype=np.uint64), np.array([int(x) for x in tx], dtype=np.uint32)

    for nf in (4, 5):   # global taxids only; then the per-record file among them
        cli("union", "-s", "--data-dir", d + "/tax", *fs[:nf], "-o", d + "/u")
        gk, gt = view(d + "/u.unik")
        ok, ot = O.union(files[:nf], taxs[:nf], tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), nf
        cli("inter", "--data-dir", d + "/tax", *fs[:nf], "-o", d + "/i")
        gk, gt = view(d + "/i.unik")
        ok, ot = O.inter(files[:nf], taxs[:nf], tax)
        assert len(ok) > 0 and np.array_equal(gk, ok) and np.array_equal(gt, ot), nf
        cli("diff", "-s", "-t", "--data-dir", d + "/tax", *fs[:nf], "-o", d + "/d")
        gk, gt = view(d + "/d.unik")
        ok, ot = O.diff(files[:nf], taxs[:nf], tax, compare_taxid=True)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), nf
        cli("diff", "-s", "--data-dir", d + "/tax", *fs[:nf], "-o", d + "/d2")
        gk, gt = view(d + "/d2.unik")
        ok, ot = O.diff(files[:nf], taxs[:nf], tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), nf
        for extra in ([], ["-n", 2]):
            cli("common", "--data-dir", d + "/tax", *extra, *fs[:nf], "-o", d + "/c")
            gk, gt = view(d + "/c.unik")
            ok, ot = O.common(files[:nf], 2 if extra else nf, taxs[:nf], tax)
            assert np.array_equal(gk, ok) and np.array_equal(gt, ot), (nf, extra)
        cli("merge", "-u", "--data-dir", d + "/tax", *fs[:nf], "-o", d + "/m")
        gk, gt = view(d + "/m.unik")
        ok, ot = O.union(files[:nf], taxs[:nf], tax)
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot), nf
    # the 3rd file's taxid is a merged id whose target is the 4th file's: diff -t of those two keeps everything
    cli("diff", "-s", "-t", "--data-dir", d + "/tax", fs[3], fs[2], "-o", d + "/d3")
    gk, gt = view(d + "/d3.unik")
    ok, ot = O.diff([files[3], files[2]], [taxs[3], taxs[2]], tax, compare_taxid=True)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot)
