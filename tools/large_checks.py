#!/usr/bin/env python
"""Validation at sizes beyond 2^32: sections 1-3 (sort of > 2^30 keys, set operations over > 2^32 records, sort of > 2^32
keys) are ALSO in the driver-run suite since round 6 (tests/test_gpu_fullsize.py::test_beyond_2_30_and_2_32_records);
section 4 (every window of > 2^32 bases through both window kernels) stays here (23 GB of outputs, 40 s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from unikmer_amd import lib

dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)


def xor_all(t):
    x = t
    while x.numel() > 1:
        if x.numel() & 1:
            x = torch.cat([x, torch.zeros(1, dtype=x.dtype, device=x.device)])
        h = x.numel() // 2
        x = x[:h] ^ x[h:]
    return int(x.item())


# 1. sort of 1.2e9 keys (status words are 64-bit from 2^30 keys on)
n = 1_600_000_000
A, _ = bench.gen_sets_device(n, 31, 0, bench.SEED, dev)   # strictly increasing, ~0.75 n elements
A = A[: min(A.numel(), 1_100_000_000)].contiguous()
assert A.numel() > (1 << 30)
g = torch.Generator(device=dev); g.manual_seed(5)
perm = torch.randperm(A.numel(), device=dev, generator=g)
S = A[perm].contiguous()
del perm
ctx.sort_u64(S, 62)
assert torch.equal(S, A), "sort of >2^30 keys failed"
print("sort n=%d ok (%.1f ms)" % (A.numel(), ctx.last_call_ms()))
del S, A
# 2. set ops with |A|+|B| > 2^32
n = 2_300_000_000
A, B = bench.gen_sets_device((4 * n + 2) // 3, 29, 0, bench.SEED + 3, dev)
na, nb = A.numel(), B.numel()
assert na + nb > (1 << 32)
out = torch.empty(na + nb, dtype=torch.int64, device=dev)
U = ctx.setop2(lib.OP_UNION, A, B, out=out)
nu = U.numel(); xu = xor_all(U); su = bool((U[1:] > U[:-1]).all())
I = ctx.setop2(lib.OP_INTER, A, B, out=out)
ni = I.numel(); xi = xor_all(I); si = bool((I[1:] > I[:-1]).all())
assert nu + ni == na + nb and su and si
assert xu == xor_all(A) ^ xor_all(B) ^ xi
print("setop |A|+|B|=%d ok: union %d inter %d" % (na + nb, nu, ni))
# 3. sort of more than 2^32 records (chunk sort + device merge)
del A, B, out, U, I
torch.cuda.empty_cache()
n = (1 << 32) + 123_456_789
g = torch.Generator(device=dev); g.manual_seed(9)
K = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g)
x0 = xor_all(K)
ctx.sort_u64(K, 62)
ok = True
step = 1 << 30
for lo in range(0, n - 1, step):                          # sortedness in slices (bounds torch temporaries)
    hi = min(n, lo + step + 1)
    ok = ok and bool((K[lo + 1:hi] >= K[lo:hi - 1]).all())
assert ok and xor_all(K) == x0, "sort of > 2^32 keys failed"
print("sort n=%d ok (%.1f ms, %.2e keys/s)" % (n, ctx.last_call_ms(), n / ctx.last_call_ms() * 1e3))
# 4. every window of more than 2^32 bases: the rolling strip kernel against the prefix-word kernel (two algorithms),
#    compared slice by slice; the Scaled sketch of the same bases through both ntHash kernels
del K
torch.cuda.empty_cache()
nb = (1 << 32) + 300_000_007
chunk = 1 << 28
bases = torch.empty(nb, dtype=torch.uint8, device=dev)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for lo in range(0, nb, chunk):
    hi = min(nb, lo + chunk)
    i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED + 11))
    bases[lo:hi] = lut[(w >> (2 * (i & 31))) & 3]
    del i, w
cuts = sorted(set([0, nb, 1_000_000_007, 1_000_000_030, (1 << 32) - 5, (1 << 32) + 17, 3_333_333_333]))
off = torch.tensor(cuts, dtype=torch.int64, device=dev)
outs = []
for flag in ("1", "0"):
    ctx.set_option("win_strip", int(flag))   # (contexts read the environment once, at creation: an option, not setenv)
    o = torch.empty(nb, dtype=torch.int64, device=dev)
    r = ctx.encode_kmers(bases, off, 31, canonical=True, out=o)
    outs.append(r)
assert outs[0].numel() == outs[1].numel() > (1 << 32)
same = True
for lo in range(0, outs[0].numel(), 1 << 30):
    same = same and bool((outs[0][lo:lo + (1 << 30)] == outs[1][lo:lo + (1 << 30)]).all())
assert same, "strip window kernel != general kernel beyond 2^32 bases"
print("encode of %d bases ok: %d windows, strip kernel == general kernel" % (nb, outs[0].numel()))
del outs, o, r
ctx.set_option("win_strip", None)
torch.cuda.empty_cache()
mh = ctx.max_hash(1000)
sk = []
for flag in ("1", "0"):
    ctx.set_option("nthash_strip", int(flag))
    sk.append(ctx.nthash(bases, off, 51, canonical=True, max_hash=mh).clone())
ctx.set_option("nthash_strip", None)
assert sk[0].numel() == sk[1].numel() > 8_000_000 and bool((sk[0] == sk[1]).all())
print("Scaled sketch of %d bases ok: %d hashes, strip kernel == general kernel" % (nb, sk[0].numel()))
