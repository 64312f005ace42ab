"""GPU parity of the point-to-plane pieces through the C ABI: the reference's own known-answer
tests (test/registration/test_registration.cpp:324-374, test/test_correspondence_estimation.cpp:45-83) and the
oracle's restatement on seeded inputs.  Normal shooting picks indices: bit-exact.  The LLS transform is floating
point (fp64 sums in a different order + the same 6x6 solve): 1e-5, north_star's bar for transforms."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from pcl_amd import capi as c
    assert c.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return c


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


def reference_surface():
    xs = np.arange(-5.0, 5.0001, 0.5, dtype=np.float32)
    P, N = [], []
    for x in xs:
        for y in xs:
            z = np.float32(0.1) * x ** 2 + np.float32(0.2) * x * y - np.float32(0.3) * y + np.float32(1.0)
            n = np.array([-0.2 * x - 0.2, 0.6 * y - 0.2, 1.0], np.float32)
            P.append([x, y, z])
            N.append(n / np.linalg.norm(n))
    return np.array(P, np.float32), np.array(N, np.float32)


GT = np.array([[0.9938, 0.0988, 0.0517, 0.1], [-0.0997, 0.9949, 0.0149, -0.2], [-0.05, -0.02, 0.9986, 0.3],
               [0, 0, 0, 1]], np.float32)


def test_lls_reference_known_motion_and_oracle(capi, orc):
    P, N = reference_surface()
    Q = (P @ GT[:3, :3].T + GT[:3, 3]).astype(np.float32)
    NQ = (N @ GT[:3, :3].T).astype(np.float32)
    rc, T = capi.estimate_point_to_plane_lls(P, Q, NQ)
    assert rc == 0 and np.abs(T - GT).max() <= 1e-2  # the reference test's bar (a linearised estimator)
    To = orc.estimate_point_to_plane_lls(P, Q, NQ)
    assert np.abs(T - To).max() <= 1e-5
    assert T[3].tolist() == [0, 0, 0, 1]
    # index lists, PointNormal-like 32-byte rows, non-finite pairs skipped (:156-172)
    rng = np.random.default_rng(4)
    si = rng.permutation(len(P)).astype(np.int32)[:300]
    src8 = np.zeros((len(P), 8), np.float32)
    src8[:, :3], src8[:, 4:7] = P, N
    tgt8 = np.zeros((len(P), 8), np.float32)
    tgt8[:, :3], tgt8[:, 4:7] = Q, NQ
    tgt8[si[5], 5] = np.nan
    src8[si[9], 0] = np.inf
    lib = capi.load_library()
    import ctypes as C
    T2 = np.zeros(16, np.float32)
    rc = lib.pclmi_estimate_point_to_plane_lls(src8.ctypes.data, len(P), 32, src8.ctypes.data + 16, 32, si.ctypes.data,
                                               tgt8.ctypes.data, len(P), 32, tgt8.ctypes.data + 16, 32, si.ctypes.data,
                                               len(si), T2.ctypes.data_as(capi.c_f32p))
    assert rc == 0
    keep = np.array([i for k, i in enumerate(si) if k not in (5, 9)], np.int32)
    To2 = orc.estimate_point_to_plane_lls(P, Q, NQ, keep, keep)
    assert np.abs(T2.reshape(4, 4) - To2).max() <= 1e-5
    # degenerate geometry (all points on one plane with one normal): singular normal equations
    flat = np.zeros((50, 3), np.float32)
    flat[:, :2] = rng.random((50, 2))
    nz = np.tile(np.array([[0, 0, 1]], np.float32), (50, 1))
    rc, Td = capi.estimate_point_to_plane_lls(flat, flat, nz)
    assert rc == 6 and np.array_equal(Td, np.eye(4, dtype=np.float32))


def test_lls_large_vs_oracle(capi, orc):
    rng = np.random.default_rng(8)
    n = 300_000
    uv = (rng.random((n, 2)) * 20 - 10).astype(np.float32)
    z = (np.sin(uv[:, 0] * 0.5) * np.cos(uv[:, 1] * 0.4)).astype(np.float32)
    P = np.stack([uv[:, 0], uv[:, 1], z], axis=1).astype(np.float32)
    nrm = np.stack([-0.5 * np.cos(uv[:, 0] * 0.5) * np.cos(uv[:, 1] * 0.4),
                    0.4 * np.sin(uv[:, 0] * 0.5) * np.sin(uv[:, 1] * 0.4), np.ones(n)], axis=1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    th = np.deg2rad(0.8)
    G = np.eye(4, dtype=np.float32)
    G[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    G[:3, 3] = [0.05, -0.02, 0.03]
    Q = (P @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
    NQ = (nrm @ G[:3, :3].T).astype(np.float32)
    rc, T = capi.estimate_point_to_plane_lls(P, Q, NQ)
    To = orc.estimate_point_to_plane_lls(P, Q, NQ)
    assert rc == 0 and np.abs(T - To).max() <= 1e-5
    assert np.abs(T - G).max() < 1e-3  # small motion: the linearisation is accurate


def test_normal_shooting_reference_planes_and_oracle(capi, orc):
    a, b = [], []
    i = np.float32(0)
    while i < 10:
        z = np.float32(0)
        while z < 5:
            a.append([i, 0, z])
            b.append([i, 2, z])
            z = np.float32(z + np.float32(0.2))
        i = np.float32(i + np.float32(0.2))
    a, b = np.array(a, np.float32), np.array(b, np.float32)
    nrm = capi.normals(capi.Tree(a), a, k=5)  # as the reference test: NormalEstimation, k = 5
    assert (np.abs(nrm[:, 1]) > 0.999).all()
    tb = capi.Tree(b)
    c = capi.correspondences_normal_shooting(tb, a, nrm[:, :3], k=10)
    assert len(c) == len(a) and (c["index_query"] == c["index_match"]).all()
    assert np.array_equal(c, orc.correspondences_normal_shooting(a, nrm[:, :3], b, 10))
    # seeded clouds, bit-exact against the oracle: default and finite max_distance, index view, k > cloud
    rng = np.random.default_rng(21)
    tgt = util.uniform_cloud(20_000, 5, 0, 10)
    src = util.uniform_cloud(3_000, 6, 0, 10)
    sn = rng.normal(size=(3000, 3)).astype(np.float32)
    sn /= np.linalg.norm(sn, axis=1, keepdims=True)
    tt = capi.Tree(tgt)
    for k, md in ((10, float(np.finfo(np.float64).max)), (4, 0.02), (33, 0.5)):
        g = capi.correspondences_normal_shooting(tt, src, sn, k=k, max_distance=md)
        o = orc.correspondences_normal_shooting(src, sn, tgt, k, md)
        assert np.array_equal(g, o) and (md != 0.02 or 0 < len(g) < len(src))
    view = np.arange(0, 3000, 7, dtype=np.int32)
    g = capi.correspondences_normal_shooting(tt, src, sn, k=10, src_indices=view)
    o = orc.correspondences_normal_shooting(src, sn, tgt, 10)
    assert np.array_equal(g, o[view])
    small = capi.Tree(tgt[:6])
    g = capi.correspondences_normal_shooting(small, src[:50], sn[:50], k=10)
    assert np.array_equal(g, orc.correspondences_normal_shooting(src[:50], sn[:50], tgt[:6], 10))


def test_back_projection_vs_oracle(capi, orc):
    """CorrespondenceEstimationBackProjection (impl/correspondence_estimation_backprojection.hpp:58-107): no reference
    test covers it (parity unpinned), so the bar is the oracle's restatement - including the arg-min the reference
    carries from one source point to the next - bit-exact, through the scan over function composition."""
    rng = np.random.default_rng(31)
    tgt = util.uniform_cloud(20_000, 7, 0, 10)
    src = util.uniform_cloud(5_000, 8, 0, 10)

    def unit(n):
        v = rng.normal(size=(n, 3)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    sn, tn = unit(len(src)), unit(len(tgt))
    tt = capi.Tree(tgt)
    for k, md in ((10, float(np.finfo(np.float64).max)), (4, 0.05), (32, 0.3), (1, 1.0)):
        g = capi.correspondences_back_projection(tt, src, sn, tn, k=k, max_distance=md)
        o = orc.correspondences_back_projection(src, sn, tgt, tn, k, md)
        assert np.array_equal(g, o), (k, md)
        assert md >= 1.0 or 0 < len(g) < len(src)
    # the carried arg-min matters: scrambling the source order changes pairs of the SAME source points
    g = capi.correspondences_back_projection(tt, src, sn, tn, k=10)
    perm = rng.permutation(len(src))
    gp = capi.correspondences_back_projection(tt, src[perm], sn[perm], tn, k=10)
    back = np.empty(len(src), np.int64)
    back[perm] = np.arange(len(src))
    assert len(g) == len(gp) == len(src)
    assert (g["index_match"] != gp["index_match"][back]).any(), "the order-dependence of the reference loop is gone"
    assert np.array_equal(gp, orc.correspondences_back_projection(src[perm], sn[perm], tgt, tn, 10))
    # index view = the reference loop over indices_ (state carried along the VIEW), non-finite source rows skipped
    view = np.arange(0, 5000, 7, dtype=np.int32)
    g = capi.correspondences_back_projection(tt, src, sn, tn, k=10, src_indices=view)
    o = orc.correspondences_back_projection(src[view], sn[view], tgt, tn, 10)
    o["index_query"] = view[o["index_query"]]
    assert np.array_equal(g, o)
    bad = src.copy()
    bad[5::50, 1] = np.nan
    g = capi.correspondences_back_projection(tt, bad, sn, tn, k=10)
    assert np.array_equal(g, orc.correspondences_back_projection(bad, sn, tgt, tn, 10)) and len(g) == len(src) - 100
    # k larger than the cloud is clamped; k > 32 is refused; missing target normals are refused
    small = capi.Tree(tgt[:6])
    g = capi.correspondences_back_projection(small, src[:50], sn[:50], tn[:6], k=10)
    assert np.array_equal(g, orc.correspondences_back_projection(src[:50], sn[:50], tgt[:6], tn[:6], 10))
    with pytest.raises(capi.PclmiError):
        capi.correspondences_back_projection(tt, src, sn, tn, k=33)
    with pytest.raises(capi.PclmiError):
        capi.correspondences_back_projection(tt, src, sn, tn[:100], k=10)
