cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo "== $t"
  if [ "$t" = base ]; then L=unikmer_amd/libunikmer_hip.so; else L=unikmer_amd/libukm_exp_$t.so; fi
  UKM_LIB_PATH=$GRAFT_REPO_ROOT/$L python - <<PY 2>/dev/null
import sys, torch, json
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
res = {}
for n in (100_000_000, 1_000_000):
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
    work = torch.empty_like(keys)
    vals = torch.arange(n, dtype=torch.int32, device=dev); wv = torch.empty_like(vals)
    ts, tp = [], []
    for _ in range(5):
        work.copy_(keys); torch.cuda.synchronize()
        ctx.sort_u64(work, 62)
        ts.append(ctx.last_call_ms())
    ok = bool((work[1:] >= work[:-1]).all())
    for _ in range(4):
        work.copy_(keys); wv.copy_(vals); torch.cuda.synchronize()
        ctx.sort_pairs(work, wv, 62)
        tp.append(ctx.last_call_ms())
    ok2 = bool((work[1:] >= work[:-1]).all()) and bool((keys[wv.long()] == work).all())
    res[n] = {"sort_ms": round(min(ts), 3), "pairs_ms": round(min(tp), 3), "ok": ok and ok2}
print(res)
PY
done
