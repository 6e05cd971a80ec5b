"""Route policy as API (include/unikmer_hip.h: ukm_ctx_set_option): a host steers a context's choice of internal routes with
options, not with the environment -- the UKM_* variables are read once, when the context is created, and an option set on
the context wins.  These contexts are created WITHOUT UKM_ENV_LIVE (the production behaviour; the other test modules run
with it because they flip knobs between calls on one context)."""
import numpy as np
import pytest

from conftest import splitmix64

pytestmark = pytest.mark.gpu

SEED = 0x756E696B6D6572


def _files(nfiles, n, p):
    j = np.arange(n, dtype=np.uint64)
    U = np.cumsum(np.uint64(1) + (splitmix64(np.uint64(SEED) ^ j) & np.uint64((1 << 20) - 1)), dtype=np.uint64)
    out = []
    for f in range(nfiles):
        h = splitmix64(np.uint64(1000 * (f + 1) + 5) ^ j)
        out.append(U[(h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < p])
    return out


def test_options_steer_routes_and_environment_is_read_once(monkeypatch):
    from oracle import oracle as O
    from unikmer_amd import lib as L
    monkeypatch.delenv("UKM_ENV_LIVE", raising=False)
    monkeypatch.setenv("UKM_PUNION", "1")             # present when the context is created: its default
    ctx = L.Context(0)
    try:
        files = _files(30, 40_000, 0.5)
        exp = O.union(files)
        assert ctx.get_option("punion") == 1 and ctx.get_option("place") is None
        assert np.array_equal(ctx.union(files), exp) and ctx.last_route() == 3
        assert ctx.stat("punion_attempts") == 1
        # the environment changes AFTER the context exists: no effect on it
        monkeypatch.setenv("UKM_PUNION", "0")
        assert np.array_equal(ctx.union(files), exp) and ctx.last_route() == 3
        # an option set on the context wins over its defaults
        ctx.set_option("punion", 0)
        assert np.array_equal(ctx.union(files), exp) and ctx.last_route() == 2
        ctx.set_option("no_kway", 1)
        assert np.array_equal(ctx.union(files), exp) and ctx.last_route() == 1
        ctx.set_option("no_kway", None)
        ctx.set_option("punion", None)
        assert ctx.get_option("punion") == 1           # back to what the environment said at creation
        assert np.array_equal(ctx.union(files), exp) and ctx.last_route() == 3
        # keep-everything merge: placement on demand
        ctx.set_option("place", 1)
        many = _files(100, 8_000, 0.8)
        got = ctx.merge_k(many, mode=L.PLAIN)
        assert ctx.last_route() == 7 and np.array_equal(got, np.sort(np.concatenate(many)))
        ctx.set_option("place", 0)
        ctx.set_option("srmerge", 0)
        got = ctx.merge_k(many, mode=L.PLAIN)
        assert ctx.last_route() == 2 and np.array_equal(got, np.sort(np.concatenate(many)))
        with pytest.raises(L.UkmError):
            ctx.set_option("no_such_option", 1)
        assert ctx.stat("workspace_bytes") > 0
    finally:
        ctx.close()
    # a second context created now sees the environment as it is now
    ctx2 = L.Context(0)
    try:
        assert ctx2.get_option("punion") == 0
        assert np.array_equal(ctx2.union(files), exp) and ctx2.last_route() == 2
    finally:
        ctx2.close()
