import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import time as _time

_SESSION_T0 = _time.monotonic()


def session_elapsed_s():
    """seconds since this pytest session was set up (the driver's GPU run has a wall-clock limit: long optional cases look at this)"""
    return _time.monotonic() - _SESSION_T0


def pytest_collection_modifyitems(config, items):
    """The driver runs `pytest -m gpu -x`: the first failure ends the run.  The cases that need one GPU PER RANK (RCCL on distinct
    devices) have never executed on hardware - the development box has one GPU, where they skip - so they go LAST: on a multi-GPU
    box a failure there cannot hide the result of any test that has a record."""
    def needs_distinct_devices(item):
        return "transport=rccl" in item.nodeid or "rccl_on_distinct_devices" in item.nodeid

    def long_case(item):         # the four-minute own-size replay: behind everything that has a record, in front of the never-run cases
        return "test_gpu_cfg5_own_size" in item.nodeid
    def world_of(item):      # smallest worlds first among them: a two-GPU box gets as far as it can
        import re
        m = re.search(r"\[(\d+)-transport=rccl", item.nodeid)
        return int(m.group(1)) if m else 0
    last = sorted((i for i in items if needs_distinct_devices(i)), key=world_of)
    items[:] = [i for i in items if not needs_distinct_devices(i) and not long_case(i)] + [i for i in items if long_case(i)] + last


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
