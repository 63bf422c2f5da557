#!/usr/bin/env python3
"""Edge-case vectors of the voxelisation rule of the weak-label sampler (missing #3).

    make -C oracle && python tests/golden/make_golden_weak_label_edges.py        (needs neither the reference nor open3d)

open3d (the reference's voxel grid, gen_sem_weak_label_rand_grid.py:178-193, open3d==0.15.2) exists neither in this image
nor under /root/reference.  What CAN be pinned offline is the published algorithm (oracle/open3d_voxel_rule.c spells it
out with its sources): this script builds the cases where an implementation of that algorithm can go wrong and stores

* ``exact.*``  -- points whose voxel index follows by hand: voxel_size 0.5 / 0.25 / 2.0 and coordinates that are multiples
  of voxel_size / 2, so that every operation of the rule is exact in binary floating point.  The expectations are computed
  HERE with integer arithmetic on the multiples (no floor, no division): points exactly ON a voxel face belong to the
  upper voxel, the minimum itself sits in the middle of voxel 0, negative coordinates, a single-point cloud.
* ``near.*``   -- float32 points whose quotient (p - origin) / 0.06 lies within a few ulps of an integer, found by search,
  including ones where multiplying by 1 / voxel_size instead of dividing gives ANOTHER voxel: the expectation is the C
  restatement's (true IEEE division, the published `Vector3d / double`)."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def c_rule():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libopen3d_voxel_rule.so"))
    lib.open3d_voxel_indices.restype = ctypes.c_int
    lib.open3d_voxel_indices.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                         ctypes.c_void_p]

    def run(xyz, vs):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        out = np.empty((len(xyz), 3), dtype=np.int32)
        origin = np.empty(3, dtype=np.float64)
        assert lib.open3d_voxel_indices(xyz.ctypes.data, len(xyz), xyz.shape[1], float(vs), out.ctypes.data, origin.ctypes.data) == 0
        return out, origin
    return run


def exact_case(seed, n, vs, lo, hi):
    """Coordinates = m * vs / 2 with integer m in [lo, hi): voxel index = (m - m_min + 1) // 2 by integer arithmetic
    (origin = (m_min - 1) * vs / 2; (m - m_min + 1) * (vs / 2) / vs = (m - m_min + 1) / 2)."""
    g = np.random.RandomState(seed)
    m = g.randint(lo, hi, (n, 3)).astype(np.int64)
    xyz = (m * (vs / 2)).astype(np.float32)
    assert (xyz.astype(np.float64) == m * (vs / 2)).all()            # exactly representable
    want = ((m - m.min(0) + 1) // 2).astype(np.int32)
    return xyz, want


def near_face_case(seed, n_keep, vs, rule):
    """float32 points next to voxel faces of a 0.06 m grid."""
    g = np.random.RandomState(seed)
    base = np.float32([-41.37, -17.052, -2.31])                      # the cloud's minimum (a corner point)
    origin = base.astype(np.float64) - vs * 0.5
    pts, flips = [base], 0
    ks = g.randint(1, 2500, 200000)
    ax = g.randint(0, 3, 200000)
    for k, a in zip(ks, ax):
        target = origin[a] + k * vs                                  # a face
        p32 = np.float32(target)
        for cand in (p32, np.nextafter(p32, np.float32(np.inf)), np.nextafter(p32, np.float32(-np.inf))):
            q = (np.float64(cand) - origin[a]) / vs
            near = abs(q - round(q)) < 4 * np.spacing(q)
            q2 = (np.float64(cand) - origin[a]) * (1.0 / vs)
            flip = np.floor(q) != np.floor(q2)
            if near and (flip or len(pts) < n_keep // 2):
                p = base.copy()
                p[a] = cand
                p[(a + 1) % 3] += np.float32(g.uniform(0, 30))       # anywhere inside the cloud on the other axes
                p[(a + 2) % 3] += np.float32(g.uniform(0, 4))
                pts.append(p)
                flips += int(flip)
        if len(pts) >= n_keep and flips >= 16:
            break
    xyz = np.stack(pts).astype(np.float32)
    want, org = rule(xyz, vs)
    assert np.allclose(org, origin)
    recip = np.floor((xyz.astype(np.float64) - org) * (1.0 / vs)).astype(np.int32)
    return xyz, want, int((recip != want).any(1).sum())


def main():
    rule = c_rule()
    out = {}
    for tag, seed, n, vs, lo, hi in (("exact.a", 1, 600, 0.5, -40, 40), ("exact.b", 2, 600, 0.25, -9, 300),
                                     ("exact.c", 3, 300, 2.0, -1000, -3), ("exact.d", 4, 1, 0.5, 7, 8)):
        xyz, want = exact_case(seed, n, vs, lo, hi)
        got, _ = rule(xyz, vs)
        assert (got == want).all(), tag                              # the C restatement agrees with the hand derivation
        out[f"{tag}.xyz"], out[f"{tag}.want"], out[f"{tag}.voxel_size"] = xyz, want, np.float64(vs)
    xyz, want, n_flip = near_face_case(5, 400, 0.06, rule)
    out["near.a.xyz"], out["near.a.want"], out["near.a.voxel_size"] = xyz, want, np.float64(0.06)
    out["near.a.reciprocal_rule_differs"] = np.int64(n_flip)
    print("near-face points", len(xyz), "of which a reciprocal-multiply rule misplaces", n_flip)
    np.savez_compressed(os.path.join(HERE, "weak_label_edges.npz"), **out)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
