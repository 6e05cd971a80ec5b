import sys, torch
sys.path.insert(0, "/root/repo")
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (500_000_000, 1_000_000_000, 1_073_741_000, 1_080_000_000, 1_500_000_000):
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    ts = []
    for _ in range(2):
        w = keys.clone(); torch.cuda.synchronize()
        ctx.sort_u64(w, 62); ts.append(ctx.last_call_ms())
    ok = bool((w[1:] >= w[:-1]).all())
    print(n, [round(t, 1) for t in ts], ok, flush=True)
    del keys, w
