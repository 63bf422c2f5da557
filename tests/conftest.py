import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))          # tests/_measure.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _matrix_engine_is_the_requested_one(request):
    """`C3D_MATRIX=<engine> pytest -m gpu` claims the whole suite ran on that engine: a test that switches
    engines and forgets to switch back would silently move every later test to another one (round 2 found
    exactly that: `finally: set_matrix_precision("f32")`).  Checked before and after every GPU test."""
    want = os.environ.get("C3D_MATRIX")
    if not want or "gpu" not in request.keywords:
        yield
        return
    from coarse3d_amd import ops
    assert ops.matrix_precision_state()[0] == want, f"engine is {ops.matrix_precision_state()[0]} before the test, not {want}"
    yield
    assert ops.matrix_precision_state()[0] == want, f"the test left the engine at {ops.matrix_precision_state()[0]}, not {want}"
