"""ctypes binding of libunikmer_hip.so (C ABI: include/unikmer_hip.h).

There is NO CPU fallback: if the HIP library is missing, cannot be loaded, or no GPU is
present, calls raise.  Arrays may be numpy arrays (host) or torch tensors (host or device);
device tensors are passed by data_ptr and never copied.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# UKM_LIB_PATH: developer override to load an experimental build of the same HIP library
SO_PATH = os.environ.get("UKM_LIB_PATH") or os.path.join(_HERE, "libunikmer_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_ILLEGAL_BASE = -1, -2, -3, -4
ERR_UNSORTED, ERR_NO_TAXONOMY, ERR_CAPACITY, ERR_K, ERR_PEER = -5, -6, -7, -8, -9
PLAIN, UNIQUE, REPEATED, REPEATED_CHUNK, SINGLETON = 0, 1, 2, 3, 4
OP_UNION, OP_INTER, OP_DIFF = 0, 1, 2
F_MIX_TAXID, F_CMP_TAXID = 2, 4
F_DEVICE_STREAMS = 256   # every stream pointer of an n-way call is a device pointer (no per-pointer driver query)

# every symbol include/unikmer_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ukm_last_error", "ukm_version", "ukm_device_count", "ukm_ctx_create", "ukm_ctx_destroy",
    "ukm_ctx_set_stream", "ukm_ctx_sync", "ukm_ctx_reserve", "ukm_ctx_trim", "ukm_dev_alloc", "ukm_dev_free",
    "ukm_copy", "ukm_host_alloc", "ukm_host_free", "ukm_copy_async", "ukm_copy_fence", "ukm_copy_sync", "ukm_last_kernel_ms", "ukm_last_call_ms", "ukm_last_route", "ukm_taxonomy_load", "ukm_taxonomy_max_taxid", "ukm_lca",
    "ukm_encode_kmers", "ukm_nthash", "ukm_minimizer", "ukm_max_hash", "ukm_sort_u64", "ukm_sort_pairs",
    "ukm_unique", "ukm_merge_k", "ukm_setop2", "ukm_union", "ukm_inter", "ukm_diff",
    "ukm_common", "ukm_common_threshold", "ukm_partition_points",
    "ukm_comm_get_unique_id", "ukm_comm_init", "ukm_comm_destroy", "ukm_comm_info", "ukm_prefix_splitters",
    "ukm_shard_exchange", "ukm_shard_plan", "ukm_shard_counts", "ukm_shard_exchange_known",
    "ukm_shard_splitters", "ukm_shard_splitters_plan", "ukm_shard_counts_tax", "ukm_shard_counts_plan", "ukm_count",
    "ukm_ctx_set_option", "ukm_ctx_unset_option", "ukm_ctx_get_option", "ukm_ctx_get_stat",
    "ukm_setop2_ft", "ukm_union_ft", "ukm_inter_ft", "ukm_diff_ft", "ukm_common_ft", "ukm_merge_k_ft",
]


class UkmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libunikmer_hip error %d: %s" % (code, msg))
        self.code = code


class IllegalBaseError(UkmError):
    pass


class UnsortedError(UkmError):
    pass


class CapacityError(UkmError):
    pass


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7).  If ours were loaded from /opt/rocm first, a later `import torch` would
    bring up a second runtime that sees no GPU.  Loading torch's copy first (without importing
    torch) makes the dynamic loader bind libunikmer_hip.so to it by SONAME."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the HIP library; raises if it has not been built (python -m unikmer_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("libunikmer_hip.so is not built (%s); run `python -m unikmer_amd.build`. "
                          "There is no CPU fallback." % SO_PATH)
    _preload_hip_runtime()
    L = C.CDLL(SO_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    pu64 = C.POINTER(C.c_uint64)
    pvp = C.POINTER(C.c_void_p)
    L.ukm_last_error.restype = C.c_char_p
    L.ukm_version.restype = i32
    L.ukm_device_count.argtypes = [C.POINTER(i32)]
    L.ukm_ctx_create.argtypes = [i32, pvp]
    L.ukm_ctx_destroy.argtypes = [vp]
    L.ukm_ctx_set_stream.argtypes = [vp, vp]
    L.ukm_ctx_sync.argtypes = [vp]
    L.ukm_ctx_reserve.argtypes = [vp, u64]
    L.ukm_ctx_trim.argtypes = [vp]
    L.ukm_dev_alloc.argtypes = [vp, u64, pvp]
    L.ukm_dev_free.argtypes = [vp, vp]
    L.ukm_copy.argtypes = [vp, vp, vp, u64]
    L.ukm_host_alloc.argtypes = [vp, u64, pvp]
    L.ukm_host_free.argtypes = [vp, vp]
    L.ukm_copy_async.argtypes = [vp, vp, vp, u64]
    L.ukm_copy_fence.argtypes = [vp]
    L.ukm_copy_sync.argtypes = [vp]
    L.ukm_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.ukm_last_call_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.ukm_last_route.argtypes = [vp]
    L.ukm_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.ukm_ctx_unset_option.argtypes = [vp, C.c_char_p]
    L.ukm_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(i32)]
    L.ukm_ctx_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(C.c_ulonglong)]
    L.ukm_taxonomy_load.argtypes = [vp, vp, vp, u64, vp, vp, u64]
    L.ukm_taxonomy_max_taxid.argtypes = [vp, C.POINTER(u32)]
    L.ukm_lca.argtypes = [vp, vp, vp, u64, vp]
    L.ukm_encode_kmers.argtypes = [vp, vp, vp, u64, i32, i32, i32, vp, u64, pu64]
    L.ukm_nthash.argtypes = [vp, vp, vp, u64, i32, i32, i32, u64, vp, u64, pu64]
    L.ukm_count.argtypes = [vp, vp, vp, u64, i32, i32, i32, i32, u64, i32, vp, u64, pu64]
    L.ukm_minimizer.argtypes = [vp, vp, vp, u64, i32, i32, i32, u64, vp, vp, u64, pu64]
    L.ukm_max_hash.argtypes = [u64]
    L.ukm_max_hash.restype = u64
    L.ukm_sort_u64.argtypes = [vp, vp, u64, i32]
    L.ukm_sort_pairs.argtypes = [vp, vp, vp, u64, i32]
    L.ukm_unique.argtypes = [vp, vp, vp, u64, i32, vp, vp, u64, pu64]
    L.ukm_merge_k.argtypes = [vp, pvp, pvp, pu64, i32, i32, i32, vp, vp, u64, pu64]
    L.ukm_setop2.argtypes = [vp, i32, vp, vp, u64, vp, vp, u64, u32, vp, vp, u64, pu64]
    for f in (L.ukm_union, L.ukm_inter):
        f.argtypes = [vp, pvp, pvp, pu64, i32, u32, vp, vp, u64, pu64]
    L.ukm_diff.argtypes = [vp, pvp, pvp, pu64, i32, vp, u32, vp, vp, u64, pu64]
    L.ukm_common.argtypes = [vp, pvp, pvp, pu64, i32, u32, u32, vp, vp, u64, pu64]
    # per-FILE taxids (one value per stream: the .unik header's global taxid)
    L.ukm_merge_k_ft.argtypes = [vp, pvp, pvp, vp, pu64, i32, i32, i32, vp, vp, u64, pu64]
    L.ukm_setop2_ft.argtypes = [vp, i32, vp, vp, u32, u64, vp, vp, u32, u64, u32, vp, vp, u64, pu64]
    for f in (L.ukm_union_ft, L.ukm_inter_ft):
        f.argtypes = [vp, pvp, pvp, vp, pu64, i32, u32, vp, vp, u64, pu64]
    L.ukm_diff_ft.argtypes = [vp, pvp, pvp, vp, pu64, i32, vp, u32, vp, vp, u64, pu64]
    L.ukm_common_ft.argtypes = [vp, pvp, pvp, vp, pu64, i32, u32, u32, vp, vp, u64, pu64]
    L.ukm_common_threshold.argtypes = [u32, C.c_double, u32]
    L.ukm_common_threshold.restype = u32
    L.ukm_partition_points.argtypes = [vp, vp, u64, vp, i32, vp]
    L.ukm_comm_get_unique_id.argtypes = [vp]
    L.ukm_comm_init.argtypes = [vp, i32, i32, vp]
    L.ukm_comm_destroy.argtypes = [vp]
    L.ukm_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.ukm_prefix_splitters.argtypes = [i32, i32, vp]
    L.ukm_shard_exchange.argtypes = [vp, vp, vp, vp, vp, vp, u64, vp, pu64]
    L.ukm_shard_plan.argtypes = [i32, i32, vp, vp, pu64]
    L.ukm_shard_counts.argtypes = [vp, vp, i32, vp]
    L.ukm_shard_counts_tax.argtypes = [vp, vp, i32, vp, vp]
    L.ukm_shard_counts_plan.argtypes = [i32, i32, i32, vp, vp]
    L.ukm_shard_exchange_known.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, pu64]
    L.ukm_shard_splitters.argtypes = [vp, pvp, pu64, i32, i32, vp]
    L.ukm_shard_splitters_plan.argtypes = [i32, i32, vp, i32, vp]
    _lib = L
    return L


class StreamTable:
    """see Context.stream_table"""

    def __init__(self, args, ref):
        self.args = args
        self.ref = ref


def _check(rc):
    if rc == OK:
        return
    msg = load().ukm_last_error().decode(errors="replace")
    cls = {ERR_ILLEGAL_BASE: IllegalBaseError, ERR_UNSORTED: UnsortedError,
           ERR_CAPACITY: CapacityError}.get(rc, UkmError)
    raise cls(rc, msg)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _is_file_taxid(t):
    """a stream's taxids given as ONE number: the file's global taxid (every record carries it)"""
    return isinstance(t, (int, np.integer))


_NP2TORCH = {np.dtype(np.uint64): "int64", np.dtype(np.uint32): "int32", np.dtype(np.uint8): "uint8"}


def _ptr(x, dtype):
    """(pointer, length, keepalive) of a numpy array / torch tensor viewed as `dtype` items."""
    if x is None:
        return None, 0, None
    if _is_torch(x):
        assert x.is_contiguous(), "tensors passed to libunikmer_hip must be contiguous"
        assert x.element_size() == np.dtype(dtype).itemsize, "tensor item size mismatch"
        return x.data_ptr() if x.numel() else None, x.numel(), x
    a = np.ascontiguousarray(x, dtype=dtype)
    return (a.ctypes.data if a.size else None), a.size, a


def _empty_like_kind(ref, n, dtype):
    """allocate an output of n items where `ref` lives (device tensor -> device tensor)."""
    if _is_torch(ref):
        import torch
        return torch.empty(max(n, 1), dtype=getattr(torch, _NP2TORCH[np.dtype(dtype)]), device=ref.device)
    return np.empty(max(n, 1), dtype=dtype)


class Context:
    """One ukm_ctx: a HIP stream + device workspace.  Not thread-safe (one per thread)."""

    def __init__(self, device=0, stream=None):
        """stream: a hipStream_t handle to borrow (0 = HIP's default stream, e.g.
        torch.cuda.current_stream().cuda_stream); None = the ctx creates its own stream, in
        which case the caller must synchronise its own producers before calling."""
        L = load()
        n = C.c_int(0)
        L.ukm_device_count(C.byref(n))
        if n.value == 0:
            raise RuntimeError("libunikmer_hip: no HIP device visible; this path has no CPU fallback")
        h = C.c_void_p()
        _check(L.ukm_ctx_create(device, C.byref(h)))
        self.h = h
        self.L = L
        if stream is not None:
            _check(L.ukm_ctx_set_stream(self.h, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "h", None):
            self.L.ukm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ----
    def set_stream(self, stream):
        _check(self.L.ukm_ctx_set_stream(self.h, C.c_void_p(stream)))

    def sync(self):
        _check(self.L.ukm_ctx_sync(self.h))

    def reserve(self, nbytes):
        _check(self.L.ukm_ctx_reserve(self.h, nbytes))

    def trim(self):
        """give the device workspace back (it is kept at the size the largest call needed)"""
        _check(self.L.ukm_ctx_trim(self.h))

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(self.L.ukm_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def last_route(self):
        """which internal route answered the last n-way call (include/unikmer_hip.h: ukm_last_route)"""
        return int(self.L.ukm_last_route(self.h))

    def set_option(self, key, value):
        """route policy of this context (include/unikmer_hip.h: ukm_ctx_set_option); value None removes the override"""
        if value is None:
            _check(self.L.ukm_ctx_unset_option(self.h, key.encode()))
        else:
            _check(self.L.ukm_ctx_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v, s = C.c_longlong(), C.c_int()
        _check(self.L.ukm_ctx_get_option(self.h, key.encode(), C.byref(v), C.byref(s)))
        return int(v.value) if s.value else None

    def stat(self, key):
        v = C.c_ulonglong()
        _check(self.L.ukm_ctx_get_stat(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def last_call_ms(self):
        ms = C.c_float()
        _check(self.L.ukm_last_call_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- taxonomy ----
    def taxonomy_load(self, child, parent, merged_old=None, merged_new=None):
        pc, n, k1 = _ptr(child, np.uint32)
        pp, n2, k2 = _ptr(parent, np.uint32)
        assert n == n2
        po, m, k3 = _ptr(merged_old, np.uint32)
        pn, m2, k4 = _ptr(merged_new, np.uint32)
        assert m == m2
        _check(self.L.ukm_taxonomy_load(self.h, pc, pp, n, po, pn, m))

    def max_taxid(self):
        v = C.c_uint32()
        _check(self.L.ukm_taxonomy_max_taxid(self.h, C.byref(v)))
        return v.value

    def lca(self, a, b):
        pa, n, k1 = _ptr(a, np.uint32)
        pb, n2, k2 = _ptr(b, np.uint32)
        assert n == n2
        out = _empty_like_kind(a, n, np.uint32)
        po, _, _ = _ptr(out, np.uint32)
        _check(self.L.ukm_lca(self.h, pa, pb, n, po))
        return out[:n]

    # ---- encode / hash ----
    def _windows(self, fn, bases, rec_off, k, canonical, circular, max_hash, out):
        pb, nb, k1 = _ptr(bases, np.uint8)
        poff, noff, k2 = _ptr(rec_off, np.uint64)
        n_rec = noff - 1
        if out is None:
            if _is_torch(rec_off):
                lens = (rec_off[1:] - rec_off[:-1])
                cap = int(lens.sum().item()) if circular else int(lens.clamp(min=k - 1).sub(k - 1).sum().item())
            else:
                lens = np.diff(np.asarray(rec_off).astype(np.int64))
                cap = int(lens.sum()) if circular else int(np.maximum(lens - (k - 1), 0).sum())
            out = _empty_like_kind(bases, cap, np.uint64)
        po, cap, _ = _ptr(out, np.uint64)
        n = C.c_uint64()
        if fn == "enc":
            rc = self.L.ukm_encode_kmers(self.h, pb, poff, n_rec, k, int(canonical), int(circular), po, cap,
                                         C.byref(n))
        else:
            rc = self.L.ukm_nthash(self.h, pb, poff, n_rec, k, int(canonical), int(circular), max_hash, po, cap,
                                   C.byref(n))
        _check(rc)
        return out[: n.value]

    def encode_kmers(self, bases, rec_off, k, canonical=True, circular=False, out=None):
        return self._windows("enc", bases, rec_off, k, canonical, circular, 0, out)

    def nthash(self, bases, rec_off, k, canonical=True, circular=False, max_hash=0, out=None):
        return self._windows("nt", bases, rec_off, k, canonical, circular, max_hash, out)

    def minimizer(self, bases, rec_off, k, w, circular=False, max_hash=0, with_pos=False):
        """Minimizer sketch of every record (sketches.NewMinimizerSketch, count.go:316).  Returns the
        emitted canonical ntHash values in record/group order (and their window indices)."""
        pb, nb, _ = _ptr(bases, np.uint8)
        poff, noff, _ = _ptr(rec_off, np.uint64)
        cap = nb
        out = _empty_like_kind(bases, cap, np.uint64)
        pos = _empty_like_kind(bases, cap, np.uint64) if with_pos else None
        po, _, _ = _ptr(out, np.uint64)
        pp = _ptr(pos, np.uint64)[0] if with_pos else None
        n = C.c_uint64()
        _check(self.L.ukm_minimizer(self.h, pb, poff, noff - 1, k, w, int(circular), max_hash, po, pp, cap,
                                    C.byref(n)))
        return (out[: n.value], pos[: n.value]) if with_pos else out[: n.value]

    def count(self, bases, rec_off, k, canonical=True, circular=False, hashed=False, max_hash=0, mode=UNIQUE, out=None):
        """`unikmer count -s` in one call (count.go:285-436,581): every window -> sort -> the distinct (`mode=UNIQUE`), repeated
        (`-d`: REPEATED) or singleton (`-u`: SINGLETON) set, sorted.  The windows stay on the device."""
        pb, nb, _ = _ptr(bases, np.uint8)
        poff, noff, _ = _ptr(rec_off, np.uint64)
        if out is None:
            cap = nb + 1
            if hashed and max_hash:
                cap = min(cap, int(nb * min(1.0, 3.0 * max_hash / float(1 << 64))) + (1 << 20))
            out = _empty_like_kind(bases, cap, np.uint64)
        po, cap, _ = _ptr(out, np.uint64)
        n = C.c_uint64()
        _check(self.L.ukm_count(self.h, pb, poff, noff - 1, int(k), int(canonical), int(circular), int(hashed), int(max_hash), int(mode), po,
                                cap, C.byref(n)))
        return out[: n.value]

    def max_hash(self, scale):
        return self.L.ukm_max_hash(scale)

    # ---- sort / scans ----
    def sort_u64(self, keys, key_bits=64):
        """In place; returns keys."""
        if not _is_torch(keys):
            assert isinstance(keys, np.ndarray) and keys.dtype == np.uint64 and keys.flags.c_contiguous
        p, n, _ = _ptr(keys, np.uint64)
        _check(self.L.ukm_sort_u64(self.h, p, n, key_bits))
        return keys

    def sort_pairs(self, keys, taxids, key_bits=64):
        if not _is_torch(keys):
            assert isinstance(keys, np.ndarray) and keys.dtype == np.uint64 and keys.flags.c_contiguous
            assert isinstance(taxids, np.ndarray) and taxids.dtype == np.uint32 and taxids.flags.c_contiguous
        pk, n, _ = _ptr(keys, np.uint64)
        pt, n2, _ = _ptr(taxids, np.uint32)
        assert n == n2
        _check(self.L.ukm_sort_pairs(self.h, pk, pt, n, key_bits))
        return keys, taxids

    def unique(self, keys, taxids=None, mode=UNIQUE, out=None, out_taxids=None):
        pk, n, k1 = _ptr(keys, np.uint64)
        pt, _, k2 = _ptr(taxids, np.uint32)
        cap = 2 * n if mode == REPEATED_CHUNK else n
        if out is None:
            out = _empty_like_kind(keys, cap, np.uint64)
        if taxids is not None and out_taxids is None:
            out_taxids = _empty_like_kind(keys, cap, np.uint32)
        po, cap, _ = _ptr(out, np.uint64)
        pot, _, _ = _ptr(out_taxids, np.uint32)
        m = C.c_uint64()
        _check(self.L.ukm_unique(self.h, pk, pt, n, mode, po, pot, cap, C.byref(m)))
        return (out[: m.value], out_taxids[: m.value]) if taxids is not None else out[: m.value]

    # ---- set operations ----
    def setop2(self, op, a, b, a_taxids=None, b_taxids=None, flags=0, out=None, out_taxids=None):
        """a_taxids / b_taxids: an array (one taxid per record), an int (ONE taxid for the whole file: the .unik
        header's global taxid -- nothing is expanded, the kernels take the scalar) or None"""
        pa, na, k1 = _ptr(a, np.uint64)
        pb, nb, k2 = _ptr(b, np.uint64)
        fa = int(a_taxids) if _is_file_taxid(a_taxids) else 0
        fb = int(b_taxids) if _is_file_taxid(b_taxids) else 0
        pta, _, k3 = _ptr(None if _is_file_taxid(a_taxids) else a_taxids, np.uint32)
        ptb, _, k4 = _ptr(None if _is_file_taxid(b_taxids) else b_taxids, np.uint32)
        tax = (a_taxids is not None and not (_is_file_taxid(a_taxids) and fa == 0)) or \
              (b_taxids is not None and not (_is_file_taxid(b_taxids) and fb == 0))
        bound = na + nb if op == OP_UNION else na
        if out is None:
            out = _empty_like_kind(a, bound, np.uint64)
        if tax and out_taxids is None:
            out_taxids = _empty_like_kind(a, bound, np.uint32)
        po, cap, _ = _ptr(out, np.uint64)
        pot, _, _ = _ptr(out_taxids, np.uint32)
        n = C.c_uint64()
        _check(self.L.ukm_setop2_ft(self.h, op, pa, pta, fa, na, pb, ptb, fb, nb, flags, po, pot, cap, C.byref(n)))
        return (out[: n.value], out_taxids[: n.value]) if tax else out[: n.value]

    def stream_table(self, keys_list, taxids_list=None):
        """Prepared pointer / length tables of a set of streams, to be passed IN PLACE OF keys_list to union / inter / diff /
        common / merge_k any number of times (taxids_list is then ignored).  A host that keeps its decoded .unik streams on
        the device builds these arrays once per file set -- they are exactly the `keys` / `taxids` / `lens` arguments of the
        C ABI; building them from 1000 torch tensors costs this Python binding ~0.4 ms per call otherwise."""
        return StreamTable(self._nway_args(list(keys_list), taxids_list), keys_list[0] if len(keys_list) else np.empty(0, np.uint64))

    def _nway_args(self, keys_list, taxids_list):
        if isinstance(keys_list, StreamTable):
            return keys_list.args
        n = len(keys_list)
        # entries of taxids_list that are plain ints are FILE taxids (one value for every record of the stream): they go to
        # the C ABI's file_taxids[] as they are
        ft = None
        if taxids_list is not None and any(_is_file_taxid(t) for t in taxids_list):
            ft_np = np.array([int(t) if _is_file_taxid(t) else 0 for t in taxids_list], dtype=np.uint32)
            taxids_list = [None if _is_file_taxid(t) else t for t in taxids_list]
            ft = (ft_np, bool(ft_np.any()))
        tax0 = taxids_list is not None and any(t is not None for t in taxids_list)
        if n >= 64 and all(_is_torch(k) for k in keys_list) and (not tax0 or all(t is None or _is_torch(t) for t in taxids_list)):
            # many device tensors (a 1000-file fold): the tables are built with numpy, not element by element through
            # ctypes (1.5 ms of a 4 ms call went into that)
            assert all(k.is_contiguous() and k.element_size() == 8 for k in keys_list)
            lens_np = np.fromiter((k.numel() for k in keys_list), dtype=np.uint64, count=n)
            kp_np = np.fromiter((k.data_ptr() if k.numel() else 0 for k in keys_list), dtype=np.uint64, count=n)
            tp_np = None
            if tax0:
                assert all(t is None or (t.is_contiguous() and t.element_size() == 4 and t.numel() == k.numel())
                           for t, k in zip(taxids_list, keys_list))
                tp_np = np.fromiter((t.data_ptr() if (t is not None and t.numel()) else 0 for t in taxids_list), dtype=np.uint64, count=n)
            kp = (C.c_void_p * n).from_buffer(kp_np)
            tp = (C.c_void_p * n).from_buffer(tp_np) if tp_np is not None else None
            lens = (C.c_uint64 * n).from_buffer(lens_np)
            on_device = all(k.is_cuda for k in keys_list) and (not tax0 or all(t is None or t.is_cuda for t in taxids_list))
            return kp, tp, lens, n, tax0 or bool(ft and ft[1]), int(lens_np.sum()), [keys_list, taxids_list, kp_np, tp_np, lens_np, "device" if on_device else "host"], ft
        keep = []
        kp = (C.c_void_p * max(n, 1))()
        tp = (C.c_void_p * max(n, 1))()
        lens = (C.c_uint64 * max(n, 1))()
        tax = taxids_list is not None and any(t is not None for t in taxids_list)
        for i, k in enumerate(keys_list):
            p, ln, ka = _ptr(k, np.uint64)
            kp[i] = p
            lens[i] = ln
            keep.append(ka)
            if tax and taxids_list[i] is not None:
                pt, lt, kt = _ptr(taxids_list[i], np.uint32)
                assert lt == ln
                tp[i] = pt
                keep.append(kt)
        total = sum(int(lens[i]) for i in range(n))
        return kp, (tp if tax else None), lens, n, tax or bool(ft and ft[1]), total, keep, ft

    def _nway(self, which, keys_list, taxids_list, bound, extra, flags, out, out_taxids):
        kp, tp, lens, n, tax, total, keep, ft = self._nway_args(keys_list, taxids_list)
        pft = ft[0].ctypes.data if ft is not None else None
        if (len(keep) and isinstance(keep[-1], str) and keep[-1] == "device" and which != "merge"
                and not os.environ.get("UKM_PY_NO_DEVICE_FLAG")):
            flags |= F_DEVICE_STREAMS      # all inputs are CUDA tensors: the library need not classify 2 x n pointers
        ref = (keys_list.ref if isinstance(keys_list, StreamTable) else keys_list[0]) if n else np.empty(0, np.uint64)
        cap = bound(total, int(lens[0]) if n else 0)
        if out is None:
            out = _empty_like_kind(ref, cap, np.uint64)
        if tax and out_taxids is None:
            out_taxids = _empty_like_kind(ref, cap, np.uint32)
        po, cap, _ = _ptr(out, np.uint64)
        pot, _, _ = _ptr(out_taxids, np.uint32)
        m = C.c_uint64()
        kpp = C.cast(kp, C.POINTER(C.c_void_p))
        tpp = C.cast(tp, C.POINTER(C.c_void_p)) if tp is not None else None
        L = self.L
        if which == "union":
            rc = L.ukm_union_ft(self.h, kpp, tpp, pft, lens, n, flags, po, pot, cap, C.byref(m))
        elif which == "inter":
            rc = L.ukm_inter_ft(self.h, kpp, tpp, pft, lens, n, flags, po, pot, cap, C.byref(m))
        elif which == "diff":
            sf = extra
            psf = None
            if sf is not None:
                sf = np.ascontiguousarray(sf, dtype=np.uint8)
                psf = sf.ctypes.data
            rc = L.ukm_diff_ft(self.h, kpp, tpp, pft, lens, n, psf, flags, po, pot, cap, C.byref(m))
        elif which == "common":
            rc = L.ukm_common_ft(self.h, kpp, tpp, pft, lens, n, extra, flags, po, pot, cap, C.byref(m))
        else:
            mode, final_round = extra
            rc = L.ukm_merge_k_ft(self.h, kpp, tpp, pft, lens, n, mode, int(final_round), po, pot, cap, C.byref(m))
        _check(rc)
        return (out[: m.value], out_taxids[: m.value]) if tax else out[: m.value]

    def union(self, keys_list, taxids_list=None, out=None, out_taxids=None):
        return self._nway("union", keys_list, taxids_list, lambda t, f: t, None, 0, out, out_taxids)

    def inter(self, keys_list, taxids_list=None, mix_taxid=False, out=None, out_taxids=None):
        return self._nway("inter", keys_list, taxids_list, lambda t, f: f, None,
                          F_MIX_TAXID if mix_taxid else 0, out, out_taxids)

    def diff(self, keys_list, taxids_list=None, compare_taxid=False, sorted_flags=None, out=None,
             out_taxids=None):
        return self._nway("diff", keys_list, taxids_list, lambda t, f: f, sorted_flags,
                          F_CMP_TAXID if compare_taxid else 0, out, out_taxids)

    def common(self, keys_list, threshold, taxids_list=None, out=None, out_taxids=None):
        return self._nway("common", keys_list, taxids_list, lambda t, f: t, int(threshold), 0, out, out_taxids)

    def merge_k(self, keys_list, taxids_list=None, mode=PLAIN, final_round=True, out=None, out_taxids=None):
        return self._nway("merge", keys_list, taxids_list, lambda t, f: 2 * t, (mode, final_round), 0, out,
                          out_taxids)

    def common_threshold(self, nfiles, proportion=1.0, number=0):
        return self.L.ukm_common_threshold(nfiles, proportion, number)

    # ---- multi-GPU exchange through the C ABI (RCCL inside the library; dist.py is the torch.distributed twin) ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        _check(load().ukm_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, uid):
        assert len(uid) == 128
        _check(self.L.ukm_comm_init(self.h, nranks, rank, C.c_char_p(uid)))

    def comm_destroy(self):
        _check(self.L.ukm_comm_destroy(self.h))

    def comm_info(self):
        n, r = C.c_int(), C.c_int()
        _check(self.L.ukm_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def prefix_splitters(self, key_bits, nranks):
        sp = np.empty(nranks, dtype=np.uint64)
        _check(self.L.ukm_prefix_splitters(key_bits, nranks, sp.ctypes.data))
        return sp

    @staticmethod
    def shard_plan(nranks, rank, gathered):
        """the collective capacity decision of ukm_shard_exchange as a pure host function: gathered =
        [source rank][nranks slice sizes | out_cap of that rank].  Returns (recv_counts, n_out); raises CapacityError
        on EVERY rank when ANY rank's buffer is too small."""
        g = np.ascontiguousarray(gathered, dtype=np.uint64).reshape(nranks, nranks + 1)
        rc = np.zeros(nranks, dtype=np.uint64)
        m = C.c_uint64()
        _check(load().ukm_shard_plan(nranks, rank, g.ctypes.data, rc.ctypes.data, C.byref(m)))
        return rc, m.value

    @staticmethod
    def shard_splitters_plan(nranks, gathered, key_bits):
        """sampled splitters from the gathered words ([rank][1 + per_rank] = record count, samples): the pure host
        function behind ukm_shard_splitters, also used by unikmer_amd/dist.py.  Returns nranks + 1 Python ints."""
        g = np.ascontiguousarray(gathered, dtype=np.uint64).reshape(nranks, -1)
        out = np.zeros(nranks + 1, dtype=np.uint64)
        _check(load().ukm_shard_splitters_plan(nranks, g.shape[1] - 1, g.ctypes.data, int(key_bits), out.ctypes.data))
        sp = [int(x) for x in out]
        if key_bits < 64:
            sp[-1] = 1 << key_bits
        return sp

    def shard_splitters(self, keys_list, key_bits):
        """collective (RCCL communicator of this context): sampled splitters for the sorted files this rank holds"""
        kp, _, lens, n, _, _, keep, _ft = self._nway_args(list(keys_list), None)
        out = np.zeros(self.comm_info()[0] + 1, dtype=np.uint64)
        _check(self.L.ukm_shard_splitters(self.h, C.cast(kp, C.POINTER(C.c_void_p)), lens, n, int(key_bits), out.ctypes.data))
        sp = [int(x) for x in out]
        if key_bits < 64:
            sp[-1] = 1 << key_bits
        return sp

    def shard_counts(self, send_counts, has_taxids=None):
        """send_counts [nfiles][nranks] -> recv_counts [nfiles][nranks] (one all-gather + one host sync for all files).
        has_taxids [nfiles] (bools): which files this rank will exchange WITH taxids; the flags ride in the same gather and
        ranks that disagree about a file all raise (ukm_shard_counts_tax)."""
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        if sc.ndim == 1:
            sc = sc[None, :]
        rc = np.zeros_like(sc)
        if has_taxids is None:
            _check(self.L.ukm_shard_counts(self.h, sc.ctypes.data, sc.shape[0], rc.ctypes.data))
        else:
            ht = np.ascontiguousarray(has_taxids, dtype=np.uint8)
            assert ht.shape == (sc.shape[0],)
            _check(self.L.ukm_shard_counts_tax(self.h, sc.ctypes.data, sc.shape[0], ht.ctypes.data, rc.ctypes.data))
        return rc

    @staticmethod
    def shard_counts_plan(nranks, rank, gathered):
        """the decision of ukm_shard_counts_tax as a pure host function: gathered = [rank][nfiles * nranks sizes | nfiles
        flags (0 / 1 / 2 = not declared)] -> recv_counts [nfiles][nranks] of `rank`; raises when two ranks disagree"""
        g = np.ascontiguousarray(gathered, dtype=np.uint64)
        nfiles = g.shape[1] // (nranks + 1)
        assert g.shape == (nranks, nfiles * (nranks + 1))
        rc = np.zeros((nfiles, nranks), dtype=np.uint64)
        _check(load().ukm_shard_counts_plan(nranks, rank, nfiles, g.ctypes.data, rc.ctypes.data))
        return rc

    def shard_exchange(self, keys, send_counts, taxids=None, recv_counts=None):
        """all-to-all-v of the contiguous slices of one sorted stream; returns (keys, taxids | None, recv_counts).
        The receive sizes are learned first (ukm_shard_counts) and the output is sized from them: n_local x nranks is
        NOT a bound on what a rank receives (a rank with few local records may own a dense prefix range).  Callers
        that already hold the counts of many files (shard_counts) pass recv_counts and skip the per-file gather."""
        pk, n, k1 = _ptr(keys, np.uint64)
        pt, _, k2 = _ptr(taxids, np.uint32)
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        assert int(sc.sum()) == n
        rc = (np.ascontiguousarray(recv_counts, dtype=np.uint64) if recv_counts is not None
              else self.shard_counts(sc, [taxids is not None])[0])
        cap = max(1, int(rc.sum()))
        out = _empty_like_kind(keys, cap, np.uint64)
        out_t = _empty_like_kind(keys, cap, np.uint32) if taxids is not None else None
        po, _, _ = _ptr(out, np.uint64)
        pot, _, _ = _ptr(out_t, np.uint32)
        m = C.c_uint64()
        _check(self.L.ukm_shard_exchange_known(self.h, pk, pt, sc.ctypes.data, rc.ctypes.data, po, pot, cap, C.byref(m)))
        return out[: m.value], (out_t[: m.value] if out_t is not None else None), rc

    def partition_points(self, keys, splitters):
        pk, n, k1 = _ptr(keys, np.uint64)
        sp = np.ascontiguousarray(splitters, dtype=np.uint64)
        cuts = np.empty(len(sp), dtype=np.uint64)
        _check(self.L.ukm_partition_points(self.h, pk, n, sp.ctypes.data, len(sp), cuts.ctypes.data))
        return cuts
