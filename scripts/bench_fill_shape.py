#!/usr/bin/env python3
"""fill2d without early exit, one slice: time per sweep against the number of 64-row bands and the row length (what does the
critical path of a sweep consist of?).  usage: python scripts/bench_fill_shape.py"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
for nx, ny in ((3000, 130), (3000, 258), (3000, 514), (3000, 1026), (3000, 3000), (1000, 3000), (6000, 1026), (300, 3000)):
    h = cases.holes(1, ny, nx, seed=4, frac=0.3)
    d0 = torch.from_numpy(h).cuda()
    ts = []
    for _ in range(3):
        d = d0.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, 1, 1e-9, 1.6, 100, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(json.dumps({"nx": nx, "ny": ny, "bands": (ny - 2 + 63) // 64, "ms_per_sweep": min(ts) * 10}), flush=True)
