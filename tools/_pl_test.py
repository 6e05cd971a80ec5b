import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["UKM_PLACE"] = "1"; os.environ["UKM_PUNION"] = "1"
from conftest import splitmix64, synth_tree
from oracle import oracle as O
from unikmer_amd import lib as L
ctx = L.Context(0)
child, parent = synth_tree(5, 8); ctx.taxonomy_load(child, parent); T = len(child)
SEED = 0x756E696B6D6572
def universe(n, gap_bits=24):
    j = np.arange(n, dtype=np.uint64)
    return np.cumsum(np.uint64(1) + (splitmix64(np.uint64(SEED) ^ j) & np.uint64((1 << gap_bits) - 1)), dtype=np.uint64)
def member(n, f, p, seed):
    h = splitmix64(np.uint64(seed + 1000 * (f + 1)) ^ np.arange(n, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p
ok_all = True
for n_univ, nfiles, p in ((3000, 40, 0.7), (20000, 100, 0.5), (1500, 300, 0.8), (50000, 33, 0.6)):
    U = universe(n_univ)
    files = [U[member(len(U), f, p, 7)] for f in range(nfiles)]
    files = [f for f in files if len(f)]
    taxs = [(np.uint64(1) + splitmix64(np.uint64(SEED + 2 + i) ^ f) % np.uint64(T)).astype(np.uint32) for i, f in enumerate(files)]
    cat = np.concatenate(files); o = np.argsort(cat, kind="stable")
    ek, et = cat[o], np.concatenate(taxs)[o]
    gk, gt = ctx.merge_k(files, taxs, mode=L.PLAIN)
    r = ctx.last_route()
    good = np.array_equal(gk, ek) and np.array_equal(gt, et)
    g2 = ctx.merge_k(files, mode=L.PLAIN); r2 = ctx.last_route()
    good2 = np.array_equal(g2, ek)
    print(n_univ, nfiles, p, "route", r, good, "plain route", r2, good2, flush=True)
    ok_all &= good and good2
print("ALL OK" if ok_all else "MISMATCH")
