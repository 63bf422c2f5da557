"""bench.py attributes HIP-event times and PMC traffic to kernel INSTANCES by name; the names come from Python mirrors of the
library's dispatch (coarse3d_amd/ops.py: _conv_kernel_name, _wgrad_kernel_name, _pw3_kernel_name).  A mirror that names an
instance the library does not contain would put a kernel into the roofline that rocprofv3 never shows.  This test walks the
layer shapes of the BASELINE configs on every engine and checks every name against the kernel symbols of the built library
(no GPU: `nm -C` on libcoarse3d_hip.so lists the kernel handles with their template arguments)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "coarse3d_amd", "libcoarse3d_hip.so")


def _symbols():
    if not os.path.exists(LIB) or shutil.which("nm") is None:
        pytest.skip("library not built or nm missing")
    out = subprocess.run(["nm", "-C", LIB], capture_output=True, text=True, check=True).stdout
    names = set()
    for line in out.splitlines():
        if "_kernel<" in line and "(anonymous namespace)::" in line:
            n = line.split("(anonymous namespace)::", 1)[1]
            names.add(n[:n.rindex(">") + 1] if n.rstrip().endswith(")") else n.strip())
    return names


def _conv_cases():
    """(B, H, W, source widths, Cout, k, dil, pad) of the SalsaNext step at the BASELINE shapes + the other backbones' oddities."""
    cases = []
    for (B, H, W) in ((8, 64, 2048), (16, 32, 1024), (8, 40, 1808)):
        for lvl, c in enumerate((32, 64, 128, 256, 256)):
            h, w = H >> lvl, W >> lvl
            if h < 2:
                continue
            cases += [(B, h, w, [c], c, 3, 1, 1), (B, h, w, [c], c, 3, 2, 2), (B, h, w, [c], c, 2, 2, 1), (B, h, w, [c, c, c], c, 1, 1, 0),
                      (B, h, w, [c], 2 * c if c < 256 else c, 1, 1, 0), (B, h, w, [max(c // 2, 16)], c, 1, 1, 0)]
        cases += [(B, H // 2, W // 2, [704], 704, 1, 1, 0), (B, H // 2, W // 2, [704], 256, 1, 1, 0), (B, H // 2, W // 2, [192], 704, 1, 1, 0),
                  (B, H, W, [32], 20, 1, 1, 0), (B, H, W, [16], 32, 3, 1, 1)]
    return cases


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_conv_and_weight_gradient_name_mirrors_name_kernels_of_the_library(mode, monkeypatch):
    import torch  # noqa: F401
    from coarse3d_amd import ops
    have = _symbols()
    monkeypatch.setattr(ops, "MFMA_MODE", mode)
    missing = set()
    for (B, H, W, srcs, cout, k, dil, pad) in _conv_cases():
        taps = ops.conv_taps(k, k, dil, pad)
        halo = max(max(abs(dy), abs(dx)) for dy, dx in taps)
        for grad in (False, True):
            for bf in ((False, True) if mode == 1 else (False,)):
                planes = (mode == 2 and len(taps) > 1) or (mode == 1 and len(taps) == 9) or (mode != 0 and len(taps) == 1 and cout > 64)
                for sm in ((False, True) if (mode == 1 and bf and grad) else (False,)):
                    for plain in ((False, True) if grad else (False,)):       # (input-gradient launches: sources without an on-load transform)
                        name, _ = ops._conv_kernel_name(B, H, W, srcs, [bf] * len(srcs), cout, taps, grad, planes, sm, plain=plain)
                        if name not in have and not (sm and name.startswith("conv_pw1_kernel<8")):      # (no such instance: ops asks c3d_conv_stat_mul_supported first)
                            missing.add(name)
        for fused in ((False, True) if mode == 2 else (False,)):
            for raw in ((False, True) if mode == 1 else (False,)):
                for ci in srcs:
                    for pre in ((False, True) if fused else (False,)):        # (pre-activation affine: keeps the four + four wave form)
                        name = ops._wgrad_kernel_name(ci, cout, len(taps), halo, fused=fused, raw=raw, h=H, pre=pre)
                        if name not in have:
                            missing.add(name)
    assert not missing, sorted(missing)[:20]


def test_three_and_six_tap_mirrors(monkeypatch):
    """The column-pair-view convs of RangeNet / SqueezeSegV3 (coarse3d_amd/rangenet.py): 6 taps over 2C channels, 3 taps onto 2 Cout."""
    from coarse3d_amd import ops
    from coarse3d_amd.rangenet import DOWN_TAPS, UP_TAPS
    have = _symbols()
    missing = set()
    for mode in (0, 1, 2):
        monkeypatch.setattr(ops, "MFMA_MODE", mode)
        for (B, H, W) in ((8, 64, 1024), (2, 8, 32), (1, 4, 16), (1, 2, 8)):
            for c in (32, 64, 256):
                for taps, ci, co in ((DOWN_TAPS, 2 * c, 2 * c), (UP_TAPS, 2 * c, 2 * c)):
                    for grad in (False, True):
                        name, _ = ops._conv_kernel_name(B, H, W, [ci], [False], co, taps, grad, mode == 2, False)
                        if name not in have:
                            missing.add(name)
                    name = ops._wgrad_kernel_name(ci, co, len(taps), 1, h=H)
                    if name not in have:
                        missing.add(name)
    assert not missing, sorted(missing)[:20]
