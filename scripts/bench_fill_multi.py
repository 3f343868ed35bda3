#!/usr/bin/env python3
"""fill2d without early exit on small batches: workgroups per slice and their shape (tuning build).
usage: python scripts/bench_fill_multi.py [nz ...]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.use_tuning_build(True)
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
nx = ny = 3000
h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
for nz in [int(v) for v in sys.argv[1:]] or [16, 1]:
    d0 = torch.from_numpy(np.stack([h] * nz)).cuda()
    for env in ({"FILL_MULTI": "0"}, {"FILL_MULTI_WAVES": "4", "FILL_MULTI_CH": "16"}, {"FILL_MULTI_WAVES": "4", "FILL_MULTI_CH": "32"},
                {"FILL_MULTI_WAVES": "8", "FILL_MULTI_CH": "16"}, {"FILL_MULTI_WAVES": "8", "FILL_MULTI_CH": "32"}, {"FILL_MULTI_WAVES": "16"}):
        for k, v in env.items(): os.environ["FIMEX_AMD_" + k] = v
        ts = []
        for _ in range(3):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-9, 1.6, 100, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        for k in env: os.environ.pop("FIMEX_AMD_" + k, None)
        print(json.dumps({"nz": nz, "env": env, "seconds_min": min(ts)}), flush=True)
