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
