"""ctypes view of the CPU oracle (oracle/libukm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() — never by the product package (unikmer_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libukm_oracle.so")

PLAIN, UNIQUE, REPEATED, REPEATED_CHUNK, SINGLETON = 0, 1, 2, 3, 4
F_TAXID, F_MIX_TAXID, F_CMP_TAXID = 1, 2, 4

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "ukm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ukmo_encode.argtypes = [_u8p, C.c_int, _u64p]
        L.ukmo_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.ukmo_revcomp.restype = C.c_uint64
        L.ukmo_canonical.argtypes = [C.c_uint64, C.c_int]
        L.ukmo_canonical.restype = C.c_uint64
        L.ukmo_decode.argtypes = [C.c_uint64, C.c_int, _u8p]
        for f in (L.ukmo_kmer_iter, L.ukmo_hash_iter):
            f.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p]
            f.restype = C.c_int64
        L.ukmo_minimizer.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p, _u64p]
        L.ukmo_minimizer.restype = C.c_int64
        L.ukmo_nthash_kmer.argtypes = [_u8p, C.c_int, _u64p, _u64p]
        L.ukmo_max_hash.argtypes = [C.c_uint64]
        L.ukmo_max_hash.restype = C.c_uint64
        L.ukmo_count_windows.argtypes = [_u8p, _u64p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_uint64, _u64p]
        L.ukmo_count_windows.restype = C.c_int64
        L.ukmo_sort_u64.argtypes = [_u64p, C.c_uint64]
        L.ukmo_sort_pairs.argtypes = [_u64p, _u32p, C.c_uint64]
        L.ukmo_tax_create.argtypes = [_u32p, _u32p, C.c_uint64, _u32p, _u32p, C.c_uint64]
        L.ukmo_tax_create.restype = C.c_void_p
        L.ukmo_tax_destroy.argtypes = [C.c_void_p]
        L.ukmo_lca.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.ukmo_lca.restype = C.c_uint32
        L.ukmo_unique.argtypes = [_u64p, _u32p, C.c_uint64, C.c_int, C.c_void_p, _u64p, _u32p]
        L.ukmo_unique.restype = C.c_uint64
        pp64, pp32 = C.POINTER(_u64p), C.POINTER(_u32p)
        L.ukmo_merge_k.argtypes = [pp64, pp32, _u64p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   _u64p, _u32p]
        L.ukmo_merge_k.restype = C.c_uint64
        for f in (L.ukmo_union, L.ukmo_inter):
            f.argtypes = [pp64, pp32, _u64p, C.c_int, C.c_uint32, C.c_void_p, _u64p, _u32p]
            f.restype = C.c_uint64
        L.ukmo_diff.argtypes = [pp64, pp32, _u64p, C.c_int, _u8p, C.c_uint32, C.c_void_p,
                                _u64p, _u32p]
        L.ukmo_diff.restype = C.c_uint64
        L.ukmo_common.argtypes = [pp64, pp32, _u64p, C.c_int, C.c_uint32, C.c_uint32,
                                  C.c_void_p, _u64p, _u32p]
        L.ukmo_common.restype = C.c_uint64
        L.ukmo_common_threshold.argtypes = [C.c_uint32, C.c_double, C.c_uint32]
        L.ukmo_common_threshold.restype = C.c_uint32
        for f in (L.ukmo_time_union2, L.ukmo_time_inter2):
            f.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint64, _u64p, _u64p]
            f.restype = C.c_double
        L.ukmo_time_setop2_allcores.argtypes = [C.c_int, _u64p, C.c_uint64, _u64p, C.c_uint64, _u64p, _u64p,
                                                C.POINTER(C.c_int)]
        L.ukmo_time_setop2_allcores.restype = C.c_double
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _seq(s):
    if isinstance(s, (bytes, bytearray)):
        return np.frombuffer(bytes(s), dtype=np.uint8)
    if isinstance(s, str):
        return np.frombuffer(s.encode(), dtype=np.uint8)
    return np.ascontiguousarray(s, dtype=np.uint8)


def encode(kmer):
    s = _seq(kmer)
    out = C.c_uint64()
    rc = lib().ukmo_encode(_p(s, _u8p), len(s), C.byref(out))
    if rc:
        raise ValueError("illegal base" if rc == -1 else "bad k")
    return out.value


def decode(code, k):
    out = np.zeros(k, dtype=np.uint8)
    lib().ukmo_decode(code, k, _p(out, _u8p))
    return out.tobytes().decode()


def revcomp(code, k):
    return lib().ukmo_revcomp(code, k)


def canonical(code, k):
    return lib().ukmo_canonical(code, k)


def _iter(fn, seq, k, canonical, circular):
    s = _seq(seq)
    cap = len(s) + (k if circular else 0)
    out = np.empty(max(cap, 1), dtype=np.uint64)
    n = fn(_p(s, _u8p), len(s), k, int(canonical), int(circular), _p(out, _u64p))
    if n == -1:
        raise ValueError("ErrShortSeq")
    if n < 0:
        raise ValueError("illegal base / bad k")
    return out[:n].copy()


def kmer_iter(seq, k, canonical=True, circular=False):
    return _iter(lib().ukmo_kmer_iter, seq, k, canonical, circular)


def hash_iter(seq, k, canonical=True, circular=False):
    return _iter(lib().ukmo_hash_iter, seq, k, canonical, circular)


def nthash_kmer(kmer):
    s = _seq(kmer)
    f, r = C.c_uint64(), C.c_uint64()
    lib().ukmo_nthash_kmer(_p(s, _u8p), len(s), C.byref(f), C.byref(r))
    return f.value, r.value


def minimizer(seq, k, w, circular=False):
    s = _seq(seq)
    cap = len(s) + k
    h = np.empty(cap, dtype=np.uint64)
    p = np.empty(cap, dtype=np.uint64)
    n = lib().ukmo_minimizer(_p(s, _u8p), len(s), k, w, int(circular), _p(h, _u64p), _p(p, _u64p))
    if n < 0:
        raise ValueError("minimizer failed: %d" % n)
    return h[:n].copy(), p[:n].copy()


def max_hash(scale):
    return lib().ukmo_max_hash(scale)


def count_windows(bases, rec_off, k, hashed=False, canonical=True, circular=False, max_hash=0):
    s = _seq(bases)
    off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    n_rec = len(off) - 1
    n = lib().ukmo_count_windows(_p(s, _u8p), _p(off, _u64p), n_rec, k, int(hashed),
                                 int(canonical), int(circular), max_hash, None)
    if n < 0:
        raise ValueError("count_windows failed: %d" % n)
    out = np.empty(max(n, 1), dtype=np.uint64)
    n2 = lib().ukmo_count_windows(_p(s, _u8p), _p(off, _u64p), n_rec, k, int(hashed),
                                  int(canonical), int(circular), max_hash, _p(out, _u64p))
    assert n2 == n
    return out[:n].copy()


def sort_u64(keys):
    a = np.array(keys, dtype=np.uint64)
    lib().ukmo_sort_u64(_p(a, _u64p), len(a))
    return a


def sort_pairs(keys, taxids):
    a = np.array(keys, dtype=np.uint64)
    t = np.array(taxids, dtype=np.uint32)
    lib().ukmo_sort_pairs(_p(a, _u64p), _p(t, _u32p), len(a))
    return a, t


class Taxonomy:
    def __init__(self, child, parent, merged_old=None, merged_new=None):
        c = np.ascontiguousarray(child, dtype=np.uint32)
        p = np.ascontiguousarray(parent, dtype=np.uint32)
        mo = np.ascontiguousarray(merged_old if merged_old is not None else [], dtype=np.uint32)
        mn = np.ascontiguousarray(merged_new if merged_new is not None else [], dtype=np.uint32)
        self.h = lib().ukmo_tax_create(_p(c, _u32p), _p(p, _u32p), len(c), _p(mo, _u32p),
                                       _p(mn, _u32p), len(mo))

    def lca(self, a, b):
        return lib().ukmo_lca(self.h, int(a), int(b))

    def __del__(self):
        try:
            lib().ukmo_tax_destroy(self.h)
        except Exception:
            pass


def _tax(t):
    return t.h if t is not None else None


def unique(keys, taxids=None, mode=UNIQUE, tax=None):
    a = np.ascontiguousarray(keys, dtype=np.uint64)
    t = np.ascontiguousarray(taxids, dtype=np.uint32) if taxids is not None else None
    ok = np.empty(2 * len(a) + 1, dtype=np.uint64)
    ot = np.empty(2 * len(a) + 1, dtype=np.uint32)
    n = lib().ukmo_unique(_p(a, _u64p), _p(t, _u32p), len(a), mode, _tax(tax), _p(ok, _u64p),
                          _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if t is not None else ok[:n].copy()


def _streams(keys_list, taxids_list):
    ks = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys_list]
    n = len(ks)
    kp = (_u64p * n)(*[_p(k, _u64p) for k in ks])
    lens = np.array([len(k) for k in ks], dtype=np.uint64)
    ts, tp = None, None
    if taxids_list is not None:
        ts = [None if t is None else np.ascontiguousarray(t, dtype=np.uint32) for t in taxids_list]  # None -> NULL (stream without taxids)
        tp = (_u32p * n)(*[_p(t, _u32p) for t in ts])
    return ks, ts, kp, tp, lens


def _setop_out(total, has_tax):
    ok = np.empty(total + 1, dtype=np.uint64)
    ot = np.empty(total + 1, dtype=np.uint32) if has_tax else None
    return ok, ot


def merge_k(keys_list, taxids_list=None, mode=PLAIN, final_round=True, tax=None):
    ks, ts, kp, tp, lens = _streams(keys_list, taxids_list)
    ok, ot = _setop_out(2 * int(lens.sum()), ts is not None)
    n = lib().ukmo_merge_k(kp, tp, _p(lens, _u64p), len(ks), mode, int(final_round), _tax(tax),
                           _p(ok, _u64p), _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if ts is not None else ok[:n].copy()


def union(keys_list, taxids_list=None, tax=None):
    ks, ts, kp, tp, lens = _streams(keys_list, taxids_list)
    flags = F_TAXID if ts is not None else 0
    ok, ot = _setop_out(int(lens.sum()), ts is not None)
    n = lib().ukmo_union(kp, tp, _p(lens, _u64p), len(ks), flags, _tax(tax), _p(ok, _u64p),
                         _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if ts is not None else ok[:n].copy()


def inter(keys_list, taxids_list=None, tax=None, mix_taxid=False):
    ks, ts, kp, tp, lens = _streams(keys_list, taxids_list)
    flags = (F_TAXID if ts is not None and not mix_taxid else 0) | (F_MIX_TAXID if mix_taxid else 0)
    ok, ot = _setop_out(int(lens[0]) if len(ks) else 0, ts is not None)
    n = lib().ukmo_inter(kp, tp, _p(lens, _u64p), len(ks), flags, _tax(tax), _p(ok, _u64p),
                         _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if ts is not None else ok[:n].copy()


def diff(keys_list, taxids_list=None, tax=None, compare_taxid=False, sorted_flags=None):
    ks, ts, kp, tp, lens = _streams(keys_list, taxids_list)
    flags = (F_TAXID if ts is not None else 0) | (F_CMP_TAXID if compare_taxid else 0)
    sf = np.ascontiguousarray(sorted_flags, dtype=np.uint8) if sorted_flags is not None else None
    ok, ot = _setop_out(int(lens[0]) if len(ks) else 0, ts is not None)
    n = lib().ukmo_diff(kp, tp, _p(lens, _u64p), len(ks), _p(sf, _u8p), flags, _tax(tax),
                        _p(ok, _u64p), _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if ts is not None else ok[:n].copy()


def common_threshold(nfiles, proportion=1.0, number=0):
    return lib().ukmo_common_threshold(nfiles, proportion, number)


def common(keys_list, threshold, taxids_list=None, tax=None):
    ks, ts, kp, tp, lens = _streams(keys_list, taxids_list)
    flags = F_TAXID if ts is not None else 0
    ok, ot = _setop_out(int(lens.sum()), ts is not None)
    n = lib().ukmo_common(kp, tp, _p(lens, _u64p), len(ks), threshold, flags, _tax(tax),
                          _p(ok, _u64p), _p(ot, _u32p))
    return (ok[:n].copy(), ot[:n].copy()) if ts is not None else ok[:n].copy()


def time_union2(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty(len(a) + len(b) + 1, dtype=np.uint64)
    n = C.c_uint64()
    t = lib().ukmo_time_union2(_p(a, _u64p), len(a), _p(b, _u64p), len(b), _p(out, _u64p),
                               C.byref(n))
    return t, out[: n.value]


def time_setop2_allcores(op, a, b):
    """All-cores 2-pass sorted merge; op 0 union, 1 inter, 2 diff.
    Returns (seconds, result, threads)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty(len(a) + len(b) + 1, dtype=np.uint64)
    n = C.c_uint64()
    th = C.c_int()
    t = lib().ukmo_time_setop2_allcores(op, _p(a, _u64p), len(a), _p(b, _u64p), len(b), _p(out, _u64p),
                                        C.byref(n), C.byref(th))
    return t, out[: n.value], th.value


def time_inter2(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty(len(a) + 1, dtype=np.uint64)
    n = C.c_uint64()
    t = lib().ukmo_time_inter2(_p(a, _u64p), len(a), _p(b, _u64p), len(b), _p(out, _u64p),
                               C.byref(n))
    return t, out[: n.value]
