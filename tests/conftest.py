import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built on demand here (hipcc cross-compiles without a GPU)."""
    from lambdaworks_cairo_prover_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "lambdaworks_cairo_prover_amd", "csrc")])
    return _lib.load()


@pytest.fixture(scope="session")
def hip_ctx(hip_lib):
    from lambdaworks_cairo_prover_amd import api
    ctx = api.Context(device=0)
    yield ctx
    ctx.close()
