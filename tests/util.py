"""Shared helpers for the tests: fixture loading and the reference tests' input recipes."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def read_pcd_ascii(path):
    """Minimal reader for the ASCII `.PCD v.5` fixtures (x y z floats), cf. test/bun0.pcd:1-10."""
    lines = Path(path).read_text().splitlines()
    n, start = None, None
    for i, ln in enumerate(lines):
        if ln.startswith("POINTS"):
            n = int(ln.split()[1])
        if ln.startswith("DATA"):
            assert ln.split()[1] == "ascii"
            start = i + 1
            break
    pts = np.array([[float(v) for v in ln.split()[:3]] for ln in lines[start:start + n]], dtype=np.float32)
    assert pts.shape == (n, 3)
    return pts


def bun0():
    return read_pcd_ascii(GOLDEN / "bun0.pcd")


def bun4():
    return read_pcd_ascii(GOLDEN / "bun4.pcd")


def golden(name):
    return json.loads((GOLDEN / name).read_text())


def grid_cloud():
    """11^3 grid, step 0.1 in [-0.5, 0.5], fp32 accumulation as test/kdtree/test_kdtree.cpp:65-71."""
    vals = []
    v = np.float32(-0.5)
    while v <= np.float32(0.5):
        vals.append(v)
        v = np.float32(v + np.float32(0.1))
    pts = [(x, y, z) for z in vals for y in vals for x in vals]
    return np.array(pts, dtype=np.float32)


def uniform_cloud(n, seed, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3), dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def ulp_diff(a, b):
    """|a-b| in units of fp32 ulps for non-negative finite floats."""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)
