#!/usr/bin/env python3
"""Fixed (per call, sweep-independent) cost of fill2d: the two scan-order sums.  Knob: FIMEX_AMD_SUM_ALGO (0 chain, 1 binade-parallel in one workgroup, 2 over the whole chip)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
nx, ny = 3000, 3000
h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
d0 = torch.from_numpy(h[None]).cuda().repeat(4, 1, 1).contiguous()
for lanes in sys.argv[1:] or ["0", "1", "2"]:
    os.environ["FIMEX_AMD_SUM_ALGO"] = lanes
    best = 1e9
    for rep in range(3):
        d = d0.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, 4, 1e-12, 1.6, 1, st); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"sum_algo": int(lanes), "ms_one_sweep_call": best * 1e3}), flush=True)
x = d0[0].contiguous().view(-1)
for algo in (0, 1, 2):
    for mode, avg in ((0, 0.0), (1, 280.0)):
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fa.scan_sum_device(x.data_ptr(), x.numel(), mode, avg, algo, st)
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"scan_sum_algo": algo, "mode": mode, "n": x.numel(), "ms": best * 1e3}), flush=True)
