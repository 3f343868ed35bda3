"""The measurement harness itself (CPU only, no product code involved): the synthetic workloads have the proportions SURVEY.md 8d
asks for, and the scripts that pick the timed launches out of a rocprofv3 trace pick the right ones."""
import importlib.util
import os

import numpy as np

import workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _positions(wl):
    lon, lat = wl.target_lonlat()
    px = workloads.axis_positions_numpy(np.degrees(lon), wl.src_lon)
    py = workloads.axis_positions_numpy(np.degrees(lat), wl.src_lat)
    return px, py


def test_c2_variants_have_the_stated_proportions():
    """default: round 1's axes (about 11 % of the targets outside the source, bounding box = the whole source);
    one_percent: SURVEY 8d's proportions (about 1 % outside, reduced-domain bounding box >= 95 % of the source)."""
    for variant, undef_lo, undef_hi, bbox_lo in (("default", 0.10, 0.12, 0.99), ("one_percent", 0.008, 0.013, 0.95)):
        wl = workloads.BilinearRotatedPole(scale=4, variant=variant)
        px, py = _positions(wl)
        inside = (px >= 0) & (px <= wl.inX - 1) & (py >= 0) & (py <= wl.inY - 1)
        assert undef_lo < 1.0 - inside.mean() < undef_hi, (variant, 1.0 - inside.mean())
        bbox = workloads.reduced_domain_cells(px, py, wl.inX, wl.inY)
        assert bbox >= bbox_lo * wl.inX * wl.inY, (variant, bbox / (wl.inX * wl.inY))
        touched = workloads.touched_source_cells(px, py, wl.inX, wl.inY, 2)
        assert touched <= bbox  # the stencils touch no cell outside the box the reference would read
    base = workloads.BilinearRotatedPole(scale=4).base_field()
    assert base.dtype == np.float32 and 0.0005 < np.isnan(base).mean() < 0.002


def test_timed_launches_are_the_ones_before_the_verification():
    """scripts/collect_profiles.py: of all launches of the staged kernel in a trace of bench.py (probing, tuning, warm-up, timed,
    extras), the timed ones are the STEPS launches that directly precede the first gather-kernel launch."""
    cp = _script("collect_profiles")
    staged = "void fimex_amd::(anonymous namespace)::staged_apply2<2, 1024, 4, 5, false, 2>(fimex_amd::(anonymous namespace)::Staged2Args)"
    gather = "void fimex_amd::(anonymous namespace)::bilinear_apply<8, true>(fimex_amd::(anonymous namespace)::ApplyArgs, unsigned int const*, float const*, float const*)"
    rows, d = [], 1
    for k in range(61):  # probing + tuning + warm-up
        rows.append({"Dispatch_Id": str(d), "Kernel_Name": staged, "tag": "before"}); d += 1
    rows.append({"Dispatch_Id": str(d), "Kernel_Name": "some_torch_kernel", "tag": "other"}); d += 1
    for k in range(cp.STEPS):
        rows.append({"Dispatch_Id": str(d), "Kernel_Name": staged, "tag": "timed"}); d += 1
    rows.append({"Dispatch_Id": str(d), "Kernel_Name": gather, "tag": "verify"}); d += 1
    for k in range(25):  # extras after the verification
        rows.append({"Dispatch_Id": str(d), "Kernel_Name": staged, "tag": "after"}); d += 1
    picked = cp.timed_dispatches(list(reversed(rows)), "Kernel_Name")
    assert len(picked) == cp.STEPS and all(r["tag"] == "timed" for r in picked)


def test_counter_json_is_read_per_instance(tmp_path):
    """scripts/placement.py: the per-instance values of a counter (one record per TCC instance) of the apply kernel's dispatches,
    in dispatch order."""
    import json
    pl = _script("placement")
    doc = {"rocprofiler-sdk-tool": [{
        "counters": [{"id": {"handle": 7}, "name": "TCC_EA0_RDREQ"}],
        "kernel_symbols": [{"kernel_id": 1, "formatted_kernel_name": "void fimex_amd::staged_apply2<2>(Args)"},
                           {"kernel_id": 2, "formatted_kernel_name": "other_kernel()"}],
        "callback_records": {"counter_collection": [
            {"dispatch_data": {"dispatch_info": {"kernel_id": 1, "dispatch_id": 5}}, "records": [{"counter_id": {"handle": 7}, "value": 3.0}, {"counter_id": {"handle": 7}, "value": 4.0}]},
            {"dispatch_data": {"dispatch_info": {"kernel_id": 2, "dispatch_id": 4}}, "records": [{"counter_id": {"handle": 7}, "value": 9.0}]},
            {"dispatch_data": {"dispatch_info": {"kernel_id": 1, "dispatch_id": 3}}, "records": [{"counter_id": {"handle": 7}, "value": 1.0}, {"counter_id": {"handle": 7}, "value": 2.0}]},
        ]}}]}
    path = tmp_path / "r.json"
    path.write_text(json.dumps(doc))
    got = pl.parse_counter_json(str(path))
    assert [g["dispatch_id"] for g in got] == [3, 5]
    assert got[0]["counters"] == {"TCC_EA0_RDREQ": [1.0, 2.0]} and got[1]["counters"] == {"TCC_EA0_RDREQ": [3.0, 4.0]}
