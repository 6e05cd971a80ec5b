"""developer tool: radix sort call ms against n (random 62-bit keys; keys only and with taxids), the bucket kernel's counting
step on / off (UKM_SORT_COUNTING), one JSON object on stdout.  usage: python tools/sort_sizes.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UKM_ENV_LIVE"] = "1"
import torch
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
res = {"note": "radix sort call ms (best of 4 after a warm-up call), random 62-bit keys, one MI355X; digit_passes = UKM_SORT_COUNTING=0 "
               "(the LDS bucket kernel of rounds 3-4: six digit passes per bucket), counting = round 5's bucket kernel (one counting "
               "step over >= capacity bins + a walk inside the bins)", "keys_only_ms": {}, "with_taxids_ms": {}}
for n in (10_000_000, 20_000_000, 50_000_000, 100_000_000, 130_000_000, 200_000_000, 500_000_000, 1_000_000_000):
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    vals = torch.arange(n, dtype=torch.int32, device=dev) if n <= 200_000_000 else None
    for pairs in (False, True):
        if pairs and vals is None:
            continue
        row = {}
        for name, knob in (("digit_passes", "0"), ("counting", "1")):
            os.environ["UKM_SORT_COUNTING"] = knob
            ts = []
            for rep in range(5):
                w = keys.clone()
                v = vals.clone() if pairs else None
                torch.cuda.synchronize()
                if pairs: ctx.sort_pairs(w, v, 62)
                else: ctx.sort_u64(w, 62)
                ts.append(ctx.last_call_ms())
            assert bool((w[1:] >= w[:-1]).all())
            row[name] = round(min(ts[1:]), 3)
        res["with_taxids_ms" if pairs else "keys_only_ms"]["%g" % n] = row
    del keys, vals
print(json.dumps(res))
