"""Builds libunikmer_hip.so (gfx950) in-tree with hipcc.  `python -m unikmer_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libunikmer_hip.so")
SOURCES = ["ukm_ctx.hip", "ukm_setops.hip", "ukm_scan.hip", "ukm_sort.hip", "ukm_encode.hip",
           "ukm_tax.hip", "ukm_nway.hip", "ukm_kway.hip", "ukm_comm.hip", "ukm_fold.hip", "ukm_punion.hip", "ukm_pfold.hip", "ukm_srmerge.hip"]
HEADERS = ["ukm_internal.h", "ukm_device.h", "ukm_kway.h", "ukm_fold.h", "ukm_punion.h", "ukm_pfold.h", "ukm_srmerge.h", os.path.join("..", "..", "include", "unikmer_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        text = out.decode(errors="replace")
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, text))
        elif "reserved registers" in text or "failed to meet occupancy target" in text:
            # ukm_sort.hip's hand-scheduled match-any names s80..s83 in its clobber list: if a change of register
            # budget ever makes the compiler reserve them (or miss a declared occupancy) the build must not go on
            failed = True
            first = [ln for ln in text.splitlines() if "reserved registers" in ln or "occupancy target" in ln][:3]
            sys.stderr.write("register-budget check failed for %s:\n%s\n" % (src, "\n".join(first)))
            try:
                os.remove(os.path.join(CSRC, src.replace(".hip", ".o")))
            except OSError:
                pass
        elif verbose and out:
            sys.stderr.write(text)
    if failed:
        raise RuntimeError("building libunikmer_hip.so failed")
    if force or procs or _stale(SO, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_cli(force=force, verbose=verbose)
    return SO


HOST = os.path.join(HERE, "host")
CLI = os.path.join(HERE, "bin", "unikmer")


def build_cli(force=False, verbose=False):
    """the `unikmer`-compatible C++ driver (host side of the drop-in), linked against the C ABI"""
    srcs = [os.path.join(HOST, "main.cpp"), os.path.join(HOST, "unik.hpp"),
            os.path.join(HERE, "..", "include", "unikmer_hip.h")]
    if not (force or _stale(CLI, srcs + [SO])):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", os.path.join(HOST, "main.cpp"), "-o", CLI,
           "-L" + HERE, "-lunikmer_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
