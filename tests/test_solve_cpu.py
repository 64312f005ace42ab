"""pclmi_icp_solve (host code, no GPU): the rigid transform from the 24 fp64 moments against a numpy Kabsch / SVD
solution (transformation_estimation_svd.hpp:137-168), over well-conditioned, nearly planar, exactly planar (rank 2,
the SVD path), reflected-looking and three-point inputs.  The fast path is a scaled Newton polar iteration
(rigid_solve.hpp polar_rotation); both paths must agree with the SVD answer far inside the 1e-5 bar."""
import numpy as np
import pytest


def sums_of(src, tgt, origin):
    s = src.astype(np.float64) - origin
    t = tgt.astype(np.float64) - origin
    h = np.zeros(24)
    n = len(s)
    h[0] = n
    h[1:4] = s.sum(0)
    h[4:7] = t.sum(0)
    h[7:16] = (s.T @ t).reshape(9)
    h[18] = n
    h[19:22] = n * origin
    return h


def kabsch(src, tgt):
    s, t = src.astype(np.float64), tgt.astype(np.float64)
    cs, ct = s.mean(0), t.mean(0)
    H = (s - cs).T @ (t - ct)
    U, S, Vt = np.linalg.svd(H)
    V = Vt.T
    if np.linalg.det(U) * np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    R = V @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = ct - R @ cs
    return T


def rot(rng, angle):
    a = rng.normal(size=3)
    a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


@pytest.mark.parametrize("case", ["cloud", "flat1e-3", "flat1e-6", "planar", "line", "three", "big_angle", "noisy", "mirror"])
def test_solve_matches_svd(case):
    from pcl_amd import capi
    rng = np.random.default_rng(hash(case) % 2**32)
    for trial in range(20):
        n = 3 if case == "three" else 500
        src = rng.normal(size=(n, 3)) * 10 + rng.normal(size=3) * 100
        if case.startswith("flat"):
            src[:, 2] = src[:, 2] * float(case[4:])
        if case == "planar":
            src[:, 2] = 0.0
        if case == "line":
            src[:, 1:] *= 1e-7
        R = rot(rng, np.pi * 0.9 if case == "big_angle" else rng.uniform(0, 0.3))
        tgt = src @ R.T + rng.normal(size=3)
        if case == "noisy":
            tgt += rng.normal(size=tgt.shape) * 3.0
        if case == "mirror":  # the best orthogonal map is a reflection: the determinant fix must pick a rotation
            tgt = src * np.array([1.0, 1.0, -1.0]) + rng.normal(size=tgt.shape) * 0.01
        src32, tgt32 = src.astype(np.float32), tgt.astype(np.float32)
        T = capi.icp_solve(sums_of(src32, tgt32, tgt32[0].astype(np.float64)))
        assert T[3].tolist() == [0, 0, 0, 1]
        Rg = T[:3, :3].astype(np.float64)
        assert np.abs(Rg @ Rg.T - np.eye(3)).max() < 1e-6 and np.linalg.det(Rg) > 0.999
        if case in ("line",):
            continue  # rotation about the line is free: only orthonormality is defined
        Tk = kabsch(src32, tgt32)
        if case in ("planar", "three", "mirror", "flat1e-6"):
            # (near-)degenerate smallest singular value: compare what is defined - the image of the points
            got = src32.astype(np.float64) @ Rg.T + T[:3, 3]
            want = src32.astype(np.float64) @ Tk[:3, :3].T + Tk[:3, 3]
            assert np.abs(got - want).max() < 2e-3 * max(1.0, np.abs(want).max() / 100)
        else:
            tol = 2e-5 if case == "flat1e-3" else 2e-6
            assert np.abs(T[:3, :3] - Tk[:3, :3]).max() < tol, (case, trial)
            assert np.abs(T[:3, 3] - Tk[:3, 3]).max() < 1e-3 * (1 + tol / 2e-6)
