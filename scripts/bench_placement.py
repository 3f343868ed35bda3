#!/usr/bin/env python3
"""Does the time of the headline launch depend on where its buffers lie?  One process, one plan, both workgroup shapes; the
source and the output batch are views at varying byte offsets into two oversized allocations, and fresh allocations in between.
usage: python scripts/bench_placement.py"""
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
slack = 64 * 1024 * 1024 // 4
base = bench.make_slices(torch, wl.base_field(), nz).view(-1)

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

for trial in range(3):
    big_in = torch.empty(nin + slack, dtype=torch.float32, device="cuda")
    big_out = torch.empty(nout + slack, dtype=torch.float32, device="cuda")
    for off_in, off_out in ((0, 0), (64, 0), (0, 64), (1024, 0), (0, 1024), (16384, 0), (0, 16384), (262144, 0), (0, 262144), (4194304, 0), (0, 4194304), (1000, 3000)):
        d_in = big_in[off_in:off_in + nin]; d_in.copy_(base)
        d_out = big_out[off_out:off_out + nout]
        print(json.dumps({"trial": trial, "in_ptr": hex(d_in.data_ptr()), "out_ptr": hex(d_out.data_ptr()), "off_in_bytes": off_in * 4, "off_out_bytes": off_out * 4,
                          "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}), flush=True)
    del big_in, big_out
    junk = torch.empty((trial + 1) * 300 * 1024 * 1024, dtype=torch.float32, device="cuda")  # shifts the next allocations
    torch.cuda.empty_cache()
