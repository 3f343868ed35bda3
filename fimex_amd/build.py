"""Builds libfimex_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Usage: python -m fimex_amd.build [--force] [--jobs N]
The library travels to the GPU box with the source tree; nothing is JIT-compiled there.
"""
import argparse
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libfimex_amd.so")
TUNING_LIB = os.path.join(HERE, "libfimex_amd_tuning.so")  # the same sources with -DFIMEX_AMD_TUNING: reads FIMEX_AMD_<NAME> switches
HOSTLIB = os.path.join(HERE, "libfimex_amd_host.so")
HOSTCLI = os.path.join(HERE, "host_cli")  # executable of the C++ host mirror (listed in .gitignore by name)

DEVICE_SOURCES = ["capi.hip", "regrid.hip", "staged.hip", "staged2.hip", "forward.hip", "forward_tiled.hip", "vector.hip", "convert.hip", "fill.hip", "projection.hip", "coordsearch.hip", "hostpipe.hip", "batch.hip"]

# -ffp-contract=off: the kernels reproduce the reference's IEEE arithmetic operation by operation
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP kernels cannot be built")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "fimex_amd.h"))
    return hdrs


def _compile(src, force, tuning=False):
    obj = os.path.join(OBJ, os.path.basename(src) + (".tuning.o" if tuning else ".o"))
    if force or _newer(obj, [src] + _headers()):
        cmd = [_hipcc()] + HIPCC_FLAGS + (["-DFIMEX_AMD_TUNING"] if tuning else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_device(force=False, jobs=4, tuning=False):
    """libfimex_amd.so, or with tuning=True libfimex_amd_tuning.so (experiment switches read from the environment)."""
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in DEVICE_SOURCES]
    lib = TUNING_LIB if tuning else LIB
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, tuning), srcs))
    if force or _newer(lib, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [
            "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


def build_host(force=False):
    """C++ host mirror of the reference classes, linked against the C ABI only."""
    if not os.path.isdir(HOST):
        return None
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cc") and f not in ("host_cli.cc", "Projection.cc"))
    if not srcs:
        return None
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "fimex_amd.h")]
    if force or _newer(HOSTLIB, deps + [LIB]):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-ffp-contract=off",
               "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-o", HOSTLIB] + srcs + [
            "-L" + HERE, "-lfimex_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host library build failed:\n%s\n%s" % (r.stdout, r.stderr))
    cli_src = os.path.join(HOST, "host_cli.cc")
    if os.path.exists(cli_src) and (force or _newer(HOSTCLI, [cli_src, HOSTLIB] + deps)):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-o", HOSTCLI, cli_src,
               "-L" + HERE, "-lfimex_amd_host", "-lfimex_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host_cli build failed:\n%s\n%s" % (r.stdout, r.stderr))
    return HOSTLIB


def clean():
    """Removes every built artefact, so that the next build starts from the sources alone."""
    shutil.rmtree(OBJ, ignore_errors=True)
    for f in (LIB, TUNING_LIB, HOSTLIB, HOSTCLI, os.path.join(HERE, "host_cli.so")):
        if os.path.exists(f):
            os.remove(f)


def build_all(force=False, jobs=4):
    if force:
        clean()
    lib = build_device(force=force, jobs=jobs)
    build_device(force=force, jobs=jobs, tuning=True)
    host = build_host(force=force)
    return lib, host


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    print(build_all(force=a.force, jobs=a.jobs))
