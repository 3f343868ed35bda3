#!/usr/bin/env python3
"""Does the size class of the allocations matter (one buddy block of the VRAM manager against a chain of smaller ones)?  The
headline launch with its batches at the start of fresh allocations of exactly 2^k bytes against allocations of the batches' own
sizes, several times over with other allocations in between.
usage: python scripts/bench_placement5.py"""
import os, sys, json
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

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

GiB = 2 ** 30 // 4
for trial in range(4):
    for label, n_in, n_out in (("exact sizes", nin, nout), ("16 GiB + 4 GiB", 16 * GiB, 4 * GiB), ("16 GiB + 8 GiB", 16 * GiB, 8 * GiB), ("32 GiB + 32 GiB", 32 * GiB, 32 * GiB)):
        torch.cuda.empty_cache()
        junk = torch.empty((37 + 211 * trial) * 262144, dtype=torch.float32, device="cuda")
        a_in = torch.empty(n_in, dtype=torch.float32, device="cuda")
        a_out = torch.empty(n_out, dtype=torch.float32, device="cuda")
        d_in, d_out = a_in[:nin], a_out[:nout]
        d_in.view(nz, -1).copy_(src.expand(nz, -1))
        print(json.dumps({"trial": trial, "allocations": label, "in_ptr": hex(d_in.data_ptr()), "out_ptr": hex(d_out.data_ptr()),
                          "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}), flush=True)
        del a_in, a_out, d_in, d_out, junk
