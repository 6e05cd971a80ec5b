"""developer tool: the multi-GPU rebuild step on one GPU -- W sorted slices that arrive for one rank's range (stride-sharded
file: they interleave; offset-sharded: they are in value order) -> the rank's sorted file.  usage: rebuild_bench.py [N] [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from unikmer_amd import lib, dist as ud
dev = torch.device("cuda:0")
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A, _ = bench.gen_sets_device((4 * n + 2) // 3, 32, 0, bench.SEED, dev)
out = torch.empty(A.numel(), dtype=torch.int64, device=dev)
def wall(fn, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r
stride = torch.cat([A[r::W] for r in range(W)])
counts = [A[r::W].numel() for r in range(W)]
pieces = ud.split_by_counts(stride, counts)
ms, r = wall(lambda: ctx.merge_k(pieces, out=out))
assert torch.equal(r, A)
print("stride-sharded: keep-everything merge of %d slices, %d records: %.2f ms (route %d)" % (W, A.numel(), ms, ctx.last_route()))
ms, r = wall(lambda: ctx.union(pieces, out=out))
print("                the same through `union` (round 3's rebuild): %.2f ms" % ms)
off = ud.split_by_counts(A, [A.numel() // W] * (W - 1) + [A.numel() - (W - 1) * (A.numel() // W)])
ms, ok = wall(lambda: ud.pieces_in_value_order(A, [x.numel() for x in off]))
print("offset-sharded: slices in value order -> no merge; the check costs %.3f ms (%s)" % (ms, ok))
