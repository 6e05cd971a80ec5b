"""config 3's shape (files that are random halves of one universe) WITH taxids: call ms of `union`.
usage: python tools/c3_tax_bench.py [NFILES=100] [PER_FILE=5e7] [file|scalar|random|none] [reps=3]
file = every record of a file carries that file's taxid (k-mers of one genome: `count -t`) as an ARRAY of copies, scalar = the
same taxids handed over as ONE number per file (ukm_union_ft; the checksums of the two must agree), random = uniformly random"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from unikmer_amd import lib
from conftest import synth_tree
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
per = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "file"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
child, parent = synth_tree(7, 8); ctx.taxonomy_load(child, parent); T = len(child)
leaves = T - 8 ** 7 + 1
nu = 2 * per
j = torch.arange(nu, dtype=torch.int64, device=dev)
gaps = 1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1))
U = torch.cumsum(gaps, 0)
del gaps
files, taxs = [], []
for f in range(nfiles):
    h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
    k = U[(h & 1) == 1]
    files.append(k)
    if kind == "file":
        taxs.append(torch.full((k.numel(),), leaves + (f * 7919) % (8 ** 7), dtype=torch.int32, device=dev))
    elif kind == "scalar":
        taxs.append(int(leaves + (f * 7919) % (8 ** 7)))
    elif kind == "random":
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
del j, U, h
torch.cuda.empty_cache()
total = sum(x.numel() for x in files)
tx = taxs if kind != "none" else None
ok = torch.empty(nu + 8, dtype=torch.int64, device=dev)
ot = torch.empty(nu + 8, dtype=torch.int32, device=dev) if tx else None
kw = dict(out=ok, out_taxids=ot) if tx else dict(out=ok)
r = [None]
def f():
    r[0] = ctx.union(files, tx, **kw)
f()
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
out = r[0][0] if tx else r[0]
chk = int(out.sum().item()) ^ (int(r[0][1].to(torch.int64).sum().item()) if tx else 0)
print("files", nfiles, "records", total, "taxids", kind, "union ms", ["%.1f" % t for t in ts], "route", ctx.last_route(), "out", out.numel(),
      "checksum", chk, flush=True)
