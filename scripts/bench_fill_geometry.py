#!/usr/bin/env python3
"""fill2d: the two band geometries (16 waves x 16 columns, 8 waves x 32 columns) over batch sizes -- where FILL_WIDE_NZ belongs."""
import os, sys, time, json
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
for nz in (8, 16, 24, 32, 48, 64, 96):
    d0 = torch.from_numpy(h[None]).cuda().repeat(nz, 1, 1).contiguous()
    row = {"nz": nz}
    for geom, name in (("1", "16x16"), ("2", "8x32")):
        os.environ["FIMEX_AMD_FILL_GEOMETRY"] = geom
        res = {}
        for loops in (6, 16):
            best = 1e9
            for _ in range(2):
                d = d0.clone(); torch.cuda.synchronize()
                t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-12, 1.6, loops, st); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            res[loops] = best
        row["ms_per_sweep_" + name] = (res[16] - res[6]) / 10 * 1e3
    print(json.dumps(row), flush=True)
    del d0
