"""coarse3d_amd -- MI355X-native (gfx950) training hot path behind COARSE3D's pc_processor API.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (``include/coarse3d_hip.h``),
``_lib`` the ctypes binding, ``ops`` tensor-level wrappers, ``backbone`` the fused
SalsaNext forward/backward plan, ``pc_processor`` the mirror of the reference module API,
``trainer`` the step orchestration, ``dist`` the data-parallel exchange points.
"""
__version__ = "0.1.0"

import os as _os

# hipGraph replays on this ROCm (torch 2.10 bundles HIP 7.0.2): with the runtime's "graph packet capture" (AQL packets of
# a graph built once and re-submitted) a captured graph faults -- "Memory access fault by GPU" -- the first time it is
# replayed after ~2 000 unrelated launches on the device, e.g. a validation pass between two training epochs
# (tools/graph_staleness_probe.py; 600-1 500 intervening launches are fine, 2 000 fault, every time).  Building the
# packets at each launch instead costs ~6 ms of host time per replayed training step and NOTHING in wall time (33.2 vs 33.2 ms
# at 8x64x2048, 19.5 vs 19.5 ms at 16x32x1024: the GPU is the bound either way), so that is the default here.  The
# runtime reads the variable when it initialises: it takes effect if this package is imported before the first GPU call of the
# process; ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=1`` in the environment keeps the runtime's default.
import sys as _sys

_t = _sys.modules.get("torch")
_hip_up = bool(_t is not None and hasattr(_t, "cuda") and _t.cuda.is_initialized())
# False: the HIP runtime was already initialised when this package was imported and the variable was not "0" then -- the
# runtime keeps its faulting default for this process.  TrainStep(graph=True) refuses to capture in that case.
GRAPH_REPLAY_SAFE = not (_hip_up and _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0")
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
del _t, _hip_up
