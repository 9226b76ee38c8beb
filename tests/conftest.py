import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    from rgbd_pose_estimation_amd import _lib as L, api
    if L.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (the product has no CPU fallback)")
    made = []

    def make():
        c = api.Context(0)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()


@pytest.fixture(scope="session")
def G():
    """The committed golden vectors (tests/golden/make_golden.py): expected values as JSON, inputs as npz."""
    import json
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "golden.json")))
    g["arr"] = dict(np.load(os.path.join(here, "golden", "golden_inputs.npz")))
    return g
