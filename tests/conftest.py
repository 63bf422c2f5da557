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
    config.addinivalue_line("markers", "parity: GPU tests whose result depends on the matrix engine -- oracle / golden / float64 "
                                       "comparisons; the nested strict-fp32 run takes `-m \"gpu and parity\"`")


# GPU tests that do NOT go into the nested strict-fp32 run (tests/test_gpu_configs.py::test_parity_suite_passes_on_the_strict_fp32_engine):
# integer / byte paths that never touch a matrix engine, process / graph / allocator infrastructure, bench.py subprocesses, and
# the tests that pin their own engine.  Everything else in a test_gpu_*.py file carries the `parity` marker.
_NOT_ENGINE_PARITY = {
    "test_gpu_dp.py": None,                      # multi-process exchange infrastructure + bench.py subprocesses (the default run covers them)
    "test_gpu_projection.py": None, "test_gpu_knn.py": None, "test_gpu_weak_label.py": None, "test_gpu_metrics.py": None,
    "test_gpu_bf16_oracle.py": None, "test_gpu_bf16_storage.py": None,          # pin the bf16 engine themselves
    "test_gpu_step.py": ("test_captured_step_survives_thousands", "test_launch_by_launch_steps_do_not_wait", "test_no_usage_mode_holds",
                         "test_graphed_inference_survives", "test_eval_forward_is_graph_capturable"),
    "test_gpu_configs.py": ("test_parity_suite_passes_on_the_strict_fp32_engine", "test_bf16_matrix_mode_at_config2_size"),
    "test_gpu_conv.py": ("test_bf16_operand_mode", "test_f16x2_forward_experiment", "test_winograd_variant", "test_streaming_pointwise"),
}


def pytest_collection_modifyitems(config, items):
    import torch
    for item in items:
        if "gpu" not in item.keywords:
            continue
        fname = os.path.basename(str(item.fspath))
        excl = _NOT_ENGINE_PARITY.get(fname, ())
        if excl is None or any(item.name.startswith(pfx) for pfx in excl):
            continue
        item.add_marker(pytest.mark.parity)
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
