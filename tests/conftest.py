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
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from gftorf_amd import _lib
    _lib.load()   # raises if libgftorf_rast.so is missing: GPU tests must never fall back
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _one_schedule_per_image_size():
    """The operator keeps its per-tile schedule per camera, the camera being known by the address of its view matrix.  The
    test helpers build fresh camera tensors for every call, so for the tests the schedule is kept per image size alone
    (deterministic: whether two calls share a schedule must not depend on the allocator); the per-camera keys have their own
    test (test_gpu_parity.py::test_schedules_are_kept_per_camera)."""
    from gftorf_amd import api
    keep = api._TILE_HINTS_PER_CAMERA
    api._TILE_HINTS_PER_CAMERA = False
    yield
    api._TILE_HINTS_PER_CAMERA = keep

