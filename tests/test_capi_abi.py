"""The C-ABI library loads, exports every symbol include/fimex_amd.h declares, and refuses to compute
without a GPU (no CPU fallback).  CPU only: no compute entry point is exercised here."""
import ctypes
import os
import re

import numpy as np
import pytest

from fimex_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fimex_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fimex_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert len(declared) >= 25
    assert sorted(capi.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), "missing export: " + name


def test_header_is_plain_c(tmp_path):
    """include/fimex_amd.h compiles as C99 on its own (plain pointers and sizes only)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "fimex_amd.h"\nint main(void){return FIMEX_AMD_OK == 1 ? 0 : 1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_abi_version_and_error_channel():
    lib = capi.load()
    assert lib.fimex_amd_abi_version() == 131
    assert isinstance(lib.fimex_amd_last_error(), bytes)


def test_no_gpu_means_loud_failure():
    """Without a gfx950 device every compute entry point fails with a message -- it never falls back to the CPU."""
    if capi.device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    with pytest.raises(capi.FimexAmdError, match="no HIP device|gfx950"):
        capi.RegridPlan(capi.BILINEAR, np.zeros(4), np.zeros(4), 2, 2, 2, 2)
    with pytest.raises(capi.FimexAmdError):
        capi.VectorPlan(np.zeros(16), 2, 2)
    with pytest.raises(capi.FimexAmdError):
        capi.fill2d_host(np.full((4, 4), np.nan, np.float32), 4.0, 1.6, 10)


def test_product_does_not_touch_the_oracle():
    """Nothing under fimex_amd/ may import, link or call oracle/ (it is test infrastructure)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "fimex_amd")):
        if "_build" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b|fimex_oracle|orc_", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_product_library_has_no_experiment_switches():
    """The shipped library compiles its measured defaults in: it holds no FIMEX_AMD_<NAME> key to look up in the
    environment (the ablation switches that skip loads or stores exist in libfimex_amd_tuning.so only), and the tuning build
    exports the same ABI."""
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"FIMEX_AMD_" not in blob
    tuning = open(capi.TUNING_LIB_PATH, "rb").read()
    assert b"FIMEX_AMD_" in tuning
    lib = ctypes.CDLL(capi.TUNING_LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), "missing export in the tuning build: " + name
