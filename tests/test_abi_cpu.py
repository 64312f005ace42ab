"""CPU-side checks of the boundary: the library loads, exports every symbol include/pclmi.h declares,
fails loudly without a GPU, and its host-only math (pclmi_icp_solve) matches the oracle."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import pcl_oracle as orc
from pcl_amd import capi
from tests import util

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "pclmi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pclmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libpclmi.so does not export %s" % n
        assert n in capi.SIGNATURES, "%s has no ctypes signature in pcl_amd/capi.py" % n
    assert set(capi.SIGNATURES) <= set(names)
    assert lib.pclmi_version().decode().startswith("pclmi")


def test_no_device_fails_loudly_instead_of_falling_back():
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.PclmiError) as e:
        capi.Tree(util.uniform_cloud(100, 1))
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)
    with pytest.raises(capi.PclmiError):
        capi.transform(util.uniform_cloud(10, 1), np.eye(4, dtype=np.float32))
    with pytest.raises(capi.PclmiError):
        capi.reject_distance(np.zeros(3, capi.CORR_DTYPE), 1.0)


def numpy_sums(src, tgt, origin=(0, 0, 0)):
    """The 24 fp64 moments of pclmi_icp_step for already-paired points (include/pclmi.h layout)."""
    s = src.astype(np.float64) - origin
    t = tgt.astype(np.float64) - origin
    out = np.zeros(capi.ICP_NSUMS)
    out[0] = len(s)
    out[1:4] = s.sum(0)
    out[4:7] = t.sum(0)
    out[7:16] = (s.T @ t).reshape(9)
    out[18] = len(s)
    out[19:22] = np.asarray(origin, np.float64) * len(s)
    return out


def test_icp_solve_matches_oracle_estimator():
    api = util.golden("registration_api.json")
    src, tgt = util.bun0(), util.bun4()
    rec = orc.correspondences(src, tgt, reciprocal=True)
    s, t = src[rec["index_query"]], tgt[rec["index_match"]]
    T = capi.icp_solve(numpy_sums(s, t))
    assert np.allclose(T, np.array(api["transform_from_SVD"]), atol=1e-4)
    assert np.abs(T - orc.estimate_rigid_svd(s, t, scalar_double=True)).max() <= 1e-6
    # moments about another origin give the same transform
    T2 = capi.icp_solve(numpy_sums(s, t, origin=(0.1, -0.2, 0.3)))
    assert np.abs(T - T2).max() <= 1e-6
    # reflections are repaired by the determinant fix: planar input stays a proper rotation
    rng = np.random.default_rng(2)
    p = rng.random((50, 3)).astype(np.float32)
    p[:, 2] = 0
    Tp = capi.icp_solve(numpy_sums(p, p))
    assert np.allclose(Tp, np.eye(4), atol=1e-6) and np.linalg.det(Tp[:3, :3].astype(np.float64)) > 0.999
    with pytest.raises(capi.PclmiError):
        capi.icp_solve(np.zeros(capi.ICP_NSUMS))


def test_default_params_match_reference_defaults():
    p = capi.icp_default_params()  # registration.h:88-108, icp.h:104
    assert (p.max_iterations, p.ransac_iterations, p.min_correspondences) == (10, 1000, 3)
    assert p.transformation_epsilon == 0.0 and p.ransac_outlier_threshold == 0.05
    assert p.euclidean_fitness_epsilon == -np.finfo(np.float64).max
    assert p.max_correspondence_distance == np.sqrt(np.finfo(np.float64).max)
    assert C.sizeof(capi.IcpParams) == 48 and C.sizeof(capi.IcpStats) == 96  # 32 + last_transform[16]


def test_point_normal_needs_no_gpu_and_matches_the_reference_golden():
    """pclmi_point_normal is host code: test/features/test_normal_estimation.cpp:64-70 on bun0."""
    from pcl_amd import capi
    plane, curv = capi.point_normal(util.bun0())
    assert np.abs(np.abs(plane) - [0.035592, 0.369596, 0.928511, 0.0622552]).max() <= 1e-4
    assert abs(curv - 0.0693136) <= 1e-4
    empty, c0 = capi.point_normal(np.full((3, 3), np.nan, np.float32))
    assert np.isnan(empty).all() and np.isnan(c0)
