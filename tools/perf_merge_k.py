import sys, torch, time
sys.path.insert(0, "/root/repo")
import bench
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
nu = 20_000_000
j = torch.arange(nu, dtype=torch.int64, device=dev)
U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ 77) & ((1 << 32) - 1)), 0)
files = [U[(bench.splitmix64_torch(j ^ (1000 * (f + 1))) & 1) == 1] for f in range(16)]
tot = sum(x.numel() for x in files)
for mode, name in ((lib.UNIQUE, "unique"), (lib.PLAIN, "plain"), (lib.REPEATED, "repeated")):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.merge_k(files, mode=mode)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, "total", tot, "out", r.numel(), "ms", round(min(ts) * 1e3, 3))
