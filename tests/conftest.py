import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle.Oracle("f32")


@pytest.fixture(scope="session")
def orc_omp():
    """The same float32 oracle with its rollout loop spread over the host cores (for the large parity cases)."""
    from oracle import oracle
    oracle.build()
    return oracle.Oracle("f32_omp")


@pytest.fixture(scope="session")
def orc64():
    from oracle import oracle
    oracle.build()
    return oracle.Oracle("f64")


@pytest.fixture(scope="session")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from mbd_hip import _capi
    return _capi.load()


def load_model(name):
    from mbd_hip.model import Model
    path = os.path.join(ROOT, "model-based-diffusion_amd", "assets", "compiled", f"{name}.json")
    with open(path) as f:
        return Model.from_json(f.read())


@pytest.fixture()
def levers(lib):
    """Set test levers of the library for one test (include/mbd_hip_debug.h): levers(MBD_PK2=1, MBD_NO_DPP=1); -1 =
    not set.  Restored afterwards.  (The library reads the environment once; afterwards only mbd_debug_set counts.)"""
    from mbd_hip import _capi
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            if k not in saved:
                saved[k] = _capi.debug_get(k)
            _capi.debug_set(k, int(v))
    yield set_
    for k, v in saved.items():
        _capi.debug_set(k, v)
