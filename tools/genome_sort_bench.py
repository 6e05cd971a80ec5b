"""developer tool: count path (encode -> sort -> unique) on the three fixture genomes laid end to end REP times (the second and
later copies with every 97th base substituted): real low-complexity k-mers in the radix sort's top-bits buckets.
usage: genome_sort_bench.py [REP]   (UKM_SORT_LOCAL=0: all passes through HBM; UKM_SORT_DEBUG=1: route decisions)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import read_fasta_gz, MG1655, IAI39, AMUC
from unikmer_amd import lib
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
seqs, offs, at = [], [0], 0
for r in range(rep):
    for name in (MG1655, IAI39, AMUC):
        s, o = read_fasta_gz(name)
        s = s.copy()
        if r:
            s[r::97] = np.frombuffer(b"CATG", dtype=np.uint8)[(s[r::97] >> 1) & 3]
        seqs.append(s)
        offs += [at + int(b) for b in o[1:]]
        at += len(s)
seq = torch.from_numpy(np.concatenate(seqs)).to(dev)
off = torch.from_numpy(np.array(offs, dtype=np.uint64).view(np.int64)).to(dev)
codes = ctx.encode_kmers(seq, off, 31, canonical=True)
n = codes.numel()
w = torch.empty_like(codes); u = torch.empty_like(codes)
def wall(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r
def sort():
    w.copy_(codes); torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sort_u64(w, 62); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
ts = [sort() for _ in range(5)]
ms_u, r = wall(lambda: ctx.unique(w, out=u))
print("windows %d  sort %.3f ms (%.2f TB/s of the 56 B per key the top-bits route moves)  unique %.3f ms -> %d distinct" % (n, min(ts), 56 * n / min(ts) / 1e9, ms_u, r.numel()))
