"""Method-name mapping of the C++ host mirror (mifi_string_to_interpolation_method, src/interpolation.c:60-101).
CPU only (host_cli --method).  The map projections themselves run on the device (fimex_amd/csrc/projection.hip) and are
checked against oracle/proj_oracle.py in tests/test_gpu_projection.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "fimex_amd", "host_cli")


def test_method_names():
    code = lambda s: int(subprocess.check_output([CLI, "--method", s]).decode())
    assert code("bilinear") == 1 and code("nearestneighbor") == 0 and code("bicubic") == 2
    assert code("coord_nearestneighbor") == 3 and code("coord_kdtree") == 4
    assert code("forward_mean") == 6 and code("forward_undef_max") == 13
    assert code("forward_undef_min") == 9  # the reference's mapping, src/interpolation.c:97-98
    assert code("no_such_method") == -1
