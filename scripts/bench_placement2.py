#!/usr/bin/env python3
"""Which buffer's placement moves the time of the headline launch?  The source batch stays where it is while the output batch is
freed and allocated again (other allocations in between), then the other way round; then both freshly allocated several times.
usage: python scripts/bench_placement2.py"""
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
base = bench.make_slices(torch, wl.base_field(), nz).view(-1)

def timed(d_in, d_out, shape):
    os.environ["FIMEX_AMD_STAGE2_USE_ALT"] = str(shape)
    ts = []
    for r in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

def fresh(n, shift_mb):
    torch.cuda.empty_cache()
    junk = torch.empty(max(1, shift_mb) * 262144, dtype=torch.float32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    del junk
    return t

def report(what, d_in, d_out):
    print(json.dumps({"what": what, "in_ptr": hex(d_in.data_ptr()), "out_ptr": hex(d_out.data_ptr()),
                      "ms_1024_threads": timed(d_in, d_out, 0), "ms_512_threads": timed(d_in, d_out, 1)}), flush=True)

d_in = fresh(nin, 1); d_in.copy_(base)
d_out = fresh(nout, 1)
report("first", d_in, d_out)
for k in range(5):
    del d_out
    d_out = fresh(nout, 700 * (k + 1))
    report("output again", d_in, d_out)
for k in range(5):
    del d_in
    d_in = fresh(nin, 900 * (k + 1)); d_in.copy_(base)
    report("source again", d_in, d_out)
# the copy of the source itself as the input (allocated first in this process)
report("source = the first allocation of the process", base, d_out)
