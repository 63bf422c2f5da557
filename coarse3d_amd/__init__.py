"""coarse3d_amd -- MI355X-native (gfx950) training hot path behind COARSE3D's pc_processor API.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (``include/coarse3d_hip.h``),
``_lib`` the ctypes binding, ``ops`` tensor-level wrappers, ``backbone`` the fused
SalsaNext forward/backward plan, ``pc_processor`` the mirror of the reference module API,
``trainer`` the step orchestration, ``dist`` the data-parallel exchange points.
"""
__version__ = "0.1.0"
