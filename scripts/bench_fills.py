#!/usr/bin/env python3
"""Times the fill kernels on [nz][ny][nx] slices with land-mask like holes (device resident, wall time of the call).
usage: python scripts/bench_fills.py [nx ny nz]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
nx, ny, nz = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3000, 3000, 16)
fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
d0 = torch.from_numpy(np.stack([h] * nz)).cuda()
def run(name, fn, env=None):
    for k, v in (env or {}).items(): os.environ["FIMEX_AMD_" + k] = v
    ts = []
    for _ in range(3):
        d = d0.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    for k in (env or {}): os.environ.pop("FIMEX_AMD_" + k, None)
    print(json.dumps({"kernel": name, "nx": nx, "ny": ny, "nz": nz, "seconds_min": min(ts), "seconds_all": ts,
                      "Mcells_per_s": nz * nx * ny / min(ts) / 1e6}), flush=True)
run("fill2d(4,1.6,100) systolic", lambda d: fa.fill2d_device(d.data_ptr(), nx, ny, nz, 4.0, 1.6, 100, st))
run("fill2d(4,1.6,100) wavefront", lambda d: fa.fill2d_device(d.data_ptr(), nx, ny, nz, 4.0, 1.6, 100, st), {"FILL_V2": "0"})
run("fill2d(1e-9,1.6,100) systolic, no early exit", lambda d: fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-9, 1.6, 100, st))
run("creepfill2d(20,2)", lambda d: fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st))
run("creepfill2d(20,2) wavefront", lambda d: fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st), {"CREEP_V2": "0"})
