import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))          # tests/_measure.py
import coarse3d_amd  # noqa: E402,F401  (process-wide runtime defaults are set at import, before the first GPU call)


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
    """Every GPU test runs on ONE engine: the library default (bf16x3, what bench.py measures) or the one
    `C3D_MATRIX=<engine> pytest -m gpu` asks for.  A test that switches engines and forgets to switch back would
    silently move every later test to another one (round 2 found exactly that: `finally:
    set_matrix_precision("f32")`).  Checked before and after every GPU test."""
    if "gpu" not in request.keywords:
        yield
        return
    from coarse3d_amd import ops
    want = os.environ.get("C3D_MATRIX", ops.DEFAULT_MATRIX)
    assert ops.matrix_precision_state()[0] == want, f"engine is {ops.matrix_precision_state()[0]} before the test, not {want}"
    yield
    assert ops.matrix_precision_state()[0] == want, f"the test left the engine at {ops.matrix_precision_state()[0]}, not {want}"
