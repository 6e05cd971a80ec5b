import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["UKM_ENV_LIVE"] = "1"
import numpy as np
from conftest import splitmix64
from unikmer_amd import lib
from oracle import oracle as O
SEED = 0x756e696b6d6572
def _universe(n, gap_bits=24, seed=SEED):
    j = np.arange(n, dtype=np.uint64)
    gaps = np.uint64(1) + (splitmix64(np.uint64(seed) ^ j) & np.uint64((1 << gap_bits) - 1))
    return np.cumsum(gaps, dtype=np.uint64)
def _member(n, f, p, seed):
    h = splitmix64(np.uint64(seed + 1000 * (f + 1)) ^ np.arange(n, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p
ctx = lib.Context(0)
rng = np.random.default_rng(41)
os.environ["UKM_PUNION"] = sys.argv[1] if len(sys.argv) > 1 else "2"
os.environ["UKM_PUNION_DEBUG"] = "1"
U = _universe(50_000, gap_bits=40)
files = [U[_member(len(U), f, 0.6, 5)] for f in range(30)]
extra = np.sort(rng.integers(0, 1 << 63, 3000, dtype=np.uint64) * np.uint64(2) + np.uint64(1))
files[12] = np.sort(np.concatenate([files[12], extra[:2000]]))
if "a" in sys.argv[2]: files[20] = np.sort(np.concatenate([files[20], extra[1000:], np.full(3, np.uint64(2**64 - 1))]))
if "b" in sys.argv[2]: files[25] = np.sort(np.concatenate([files[25], files[25][:700]]))          # a multiset
if "c" in sys.argv[2]: files[2] = np.sort(np.concatenate([files[2], np.full(2, np.uint64(2**64 - 1))]))   # all ones inside the base set
r = ctx.union(files)
print("extras", sys.argv[2], np.array_equal(r, O.union(files)), ctx.last_route(), flush=True)
