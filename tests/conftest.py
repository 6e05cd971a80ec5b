import gzip
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library reads the UKM_* knobs of the environment ONCE per context, at ukm_ctx_create (a host steers a context with
# ukm_ctx_set_option).  The tests flip knobs with monkeypatch.setenv between calls on one module-scoped context: contexts
# created under UKM_ENV_LIVE=1 keep looking at the environment (tests/test_gpu_options.py covers the production behaviour).
os.environ.setdefault("UKM_ENV_LIVE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta_gz(name):
    """Minimal FASTA reader: returns (concatenated upper-case bases as uint8 array, rec_off)."""
    seqs = []
    cur = []
    with gzip.open(os.path.join(GOLDEN, name), "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                    cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return np.frombuffer(b"".join(seqs), dtype=np.uint8), off


_genome_cache = {}


@pytest.fixture(scope="session")
def genomes():
    def get(name):
        if name not in _genome_cache:
            _genome_cache[name] = read_fasta_gz(name)
        return _genome_cache[name]

    return get


MG1655 = "Ecoli-MG1655.fasta.gz"
IAI39 = "Ecoli-IAI39.fasta.gz"
AMUC = "A.muciniphila-ATCC_BAA-835.fasta.gz"


def splitmix64(x):
    """numpy uint64 splitmix64 finaliser (SURVEY.md §8(d) synthetic generator)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_tree(depth=7, arity=8):
    """Complete `arity`-ary tree, ids 1..T, parent[t]=(t-2)//arity+1, root 1->1."""
    T = sum(arity ** d for d in range(depth + 1))
    child = np.arange(1, T + 1, dtype=np.uint32)
    parent = ((child.astype(np.int64) - 2) // arity + 1).astype(np.uint32)
    parent[0] = 1
    return child, parent


def oracle_by_value_ranges(fn, keys_list, taxids_list=None, nranges=None, kind="set", threads=None):
    """The oracle over EVERY record of many large sorted files in seconds instead of minutes: union / inter / diff / common /
    merge act on each code independently of every other code, so the value space is cut into ranges (quantiles of the
    largest file), `fn(keys_slices, taxid_slices)` -- the oracle's single-threaded C loop behind ctypes, which releases the
    GIL -- runs on the files' slices of each range in a thread pool, and the per-range results, concatenated in range
    order, ARE the oracle's result on the whole files (still every record, still the oracle's own loops; only the order of
    evaluation changed).  kind = "inter": a file without a record in a range empties that range's result -- the oracle's
    own loop would `break` at a slice of length 0 as at an EMPTY FILE (inter.go:211-217 keeps the running result there),
    which a slice of a non-empty file is not.  Returns keys, or (keys, taxids)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthreads = threads or max(1, min(64, (os.cpu_count() or 8)))
    nranges = nranges or 2 * nthreads
    ks = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys_list]
    ts = None if taxids_list is None else [None if t is None else np.ascontiguousarray(t, dtype=np.uint32) for t in taxids_list]
    big = max(ks, key=len)
    if len(big) < 4 * nranges:
        return fn(ks, ts)
    bounds = np.unique(big[(np.arange(1, nranges) * len(big)) // nranges])        # ascending, distinct
    cuts = [np.concatenate(([0], np.searchsorted(k, bounds, side="left"), [len(k)])) for k in ks]

    def one(r):
        sl = [k[c[r]:c[r + 1]] for k, c in zip(ks, cuts)]
        tl = None if ts is None else [None if t is None else t[c[r]:c[r + 1]] for t, c in zip(ts, cuts)]
        if kind == "inter" and any(len(x) == 0 for x, k in zip(sl, ks) if len(k)):
            e = np.empty(0, dtype=np.uint64)
            return (e, np.empty(0, dtype=np.uint32)) if ts is not None else e
        return fn(sl, tl)
    with ThreadPoolExecutor(nthreads) as pool:
        parts = list(pool.map(one, range(len(bounds) + 1)))
    if ts is not None:
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    return np.concatenate(parts)
