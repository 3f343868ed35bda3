import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def tuning_build():
    """The product library compiles its experiment switches in; tests that force another kernel variant through a
    FIMEX_AMD_<NAME> switch run on libfimex_amd_tuning.so (same sources, -DFIMEX_AMD_TUNING)."""
    from fimex_amd import capi
    was = capi.use_tuning_build(True)
    yield
    capi.use_tuning_build(was)
