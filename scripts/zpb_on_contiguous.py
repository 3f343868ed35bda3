#!/usr/bin/env python3
"""Is the slowness of a physically contiguous output batch a matter of the distance between the z chunks' write streams?  One plan,
one source batch (library-placed), two output batches -- one plain, one physically contiguous (BATCH_CONTIGUOUS, tuning build) --
and the launch timed on both with z chunks of different lengths (STAGE2_ZPB: the chunks of a tile write 25 x 16 MB apart by default).
usage: python scripts/zpb_on_contiguous.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True); fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole(); nz = 200
plan, _, _ = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
src = plan.alloc_source_batch(nz, candidates=6, stream=st)
d_in = bench.make_slices(torch, wl.base_field(), nz, into=src.as_tensor())
os.environ["FIMEX_AMD_BATCH_CONTIGUOUS"] = "0"
plain = plan.alloc_batch(d_in.data_ptr(), nz, positions=6, stream=st)
os.environ["FIMEX_AMD_BATCH_CONTIGUOUS"] = "1"
contig = plan.alloc_batch(d_in.data_ptr(), nz, positions=2, stream=st)
print(json.dumps({"source_ms": src.info["msAtPosition"], "plain_ms": plain.info["msAtPosition"], "contiguous_ms": contig.info["msAtPosition"]}), flush=True)
def med(ptr):
    ts = []
    for r in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, ptr, st); b.record(); torch.cuda.synchronize()
        if r: ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for zpb in (25, 20, 23, 27, 29, 33, 40, 50):
    os.environ["FIMEX_AMD_STAGE2_ZPB"] = str(zpb)
    print(json.dumps({"zpb": zpb, "ms_plain": med(plain.data_ptr), "ms_contiguous": med(contig.data_ptr)}), flush=True)
for order in (0,):
    os.environ["FIMEX_AMD_STAGE2_ZPB"] = "50"; os.environ["FIMEX_AMD_STAGE2_ORDER"] = "0"
    print(json.dumps({"order": "chunk-major, zpb 50", "ms_plain": med(plain.data_ptr), "ms_contiguous": med(contig.data_ptr)}), flush=True)
