#!/usr/bin/env python3
"""Is the placement effect the size of the page-table fragments?  The driver's fragment for a physically contiguous block is
limited by the alignment of its VIRTUAL address (hipMalloc: 2 MiB).  The headline launch on batches from torch's allocator against
batches mapped through HIP's virtual-memory calls at virtual addresses aligned to 2 MiB, 1 GiB and 16 GiB
(scripts/calib/vmm_alloc.hip), other allocations of varying size in between.
usage: python scripts/bench_placement7.py"""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True)
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
nz = 200
plan, _, _ = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
nin, nout = nz * wl.inX * wl.inY, nz * wl.outX * wl.outY
src = bench.make_slices(torch, wl.base_field(), 1).view(1, -1)
vmm = ctypes.CDLL(os.path.join(ROOT, "scripts", "calib", "bin", "libvmm_alloc.so"))

class Block(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("size", ctypes.c_size_t), ("handle", ctypes.c_void_p)]
vmm.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Block)]
vmm.vmm_free.argtypes = [ctypes.POINTER(Block)]
vmm.vmm_granularity.restype = ctypes.c_size_t

class Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

print(json.dumps({"vmm_granularity": vmm.vmm_granularity()}), flush=True)
ref = None
for trial in range(4):
    for label, align in (("torch", 0), ("vmm 2 MiB", 2 << 20), ("vmm 1 GiB", 1 << 30), ("vmm 16 GiB", 16 << 30)):
        torch.cuda.empty_cache()
        junk = torch.empty((53 + 331 * trial) * 262144, dtype=torch.float32, device="cuda")
        blocks = []
        if align == 0:
            d_in = torch.empty(nin, dtype=torch.float32, device="cuda")
            d_out = torch.empty(nout, dtype=torch.float32, device="cuda")
        else:
            bi, bo = Block(), Block()
            rc = vmm.vmm_alloc(nin * 4, align, ctypes.byref(bi)) or vmm.vmm_alloc(nout * 4, align, ctypes.byref(bo))
            if rc != 0:
                print(json.dumps({"trial": trial, "allocations": label, "error": rc}), flush=True)
                continue
            blocks = [bi, bo]
            d_in = torch.as_tensor(Raw(bi.ptr, nin), device="cuda")
            d_out = torch.as_tensor(Raw(bo.ptr, nout), device="cuda")
        d_in.view(nz, -1).copy_(src.expand(nz, -1))
        row = {"trial": trial, "allocations": label, "in_ptr": hex(d_in.data_ptr()), "out_ptr": hex(d_out.data_ptr()),
               "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}
        chk = float(torch.nan_to_num(d_out[::1001], nan=-1.0).double().sum().item())
        ref = chk if ref is None else ref
        row["same_result"] = chk == ref
        print(json.dumps(row), flush=True)
        del d_in, d_out, junk
        torch.cuda.synchronize()
        for b in blocks: vmm.vmm_free(ctypes.byref(b))
