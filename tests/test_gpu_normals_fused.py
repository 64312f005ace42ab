"""The fused search + plane-fit kernels behind pclmi_normals (knn_kernels.hip normals_finish, radius_kernels.hip
radius_normals_*): every template instance (K = 2, 4, 8, 16, 32; dense batches -> wave-uniform walk, sparse
batches -> per-lane walk; k below the kernel's K; radius neighbourhoods that overflow the per-lane buffer and are
drained several times) against a numpy fp64 plane fit on the ORACLE's neighbour lists - the neighbour set, ties at
the k-th distance included, must be the one pcl::KdTreeFLANN returns (features/impl/feature.hpp:143-159 binds the
search, features/impl/normal_3d.hpp:46-98 fits the plane).  Tolerance 5e-5 per component (fp64 moments, fp32 result).
"""
import numpy as np
import pytest

from tests import util
from tests.test_gpu_normals import close_up_to_view_sign, knn_lists, numpy_normals, radius_lists

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


def wavy(n, seed):
    rng = np.random.default_rng(seed)
    uv = rng.random((n, 2)) * 60
    z = 4 * np.sin(uv[:, 0] * 0.15) * np.cos(uv[:, 1] * 0.1) + rng.normal(0, 0.02, n)
    return np.stack([uv[:, 0], uv[:, 1], z], axis=1).astype(np.float32)


@pytest.mark.parametrize("k", [2, 3, 4, 6, 8, 11, 16, 23, 32])
def test_knn_normals_every_kernel_instance(capi, orc, k):
    pts = wavy(60_000, 5)
    vp = (30.0, 30.0, 500.0)
    t = capi.Tree(pts)
    rng = np.random.default_rng(k)
    dense = pts[rng.choice(len(pts), 30_000, replace=False)]  # half the target: wave-uniform walk
    sparse = pts[rng.choice(len(pts), 2_000, replace=False)]   # 1/30 of it: per-lane walk
    for q in (dense, sparse):
        g = capi.normals(t, q, k=k, viewpoint=vp)
        sel = rng.choice(len(q), 1500, replace=False)
        r = numpy_normals(pts, q[sel], knn_lists(orc, pts, q[sel], k), vp)
        well = np.ones(len(sel), bool) if k >= 5 else np.zeros(len(sel), bool)  # < 5 points: plane not unique
        assert close_up_to_view_sign(g[sel], r, q[sel], vp, 5e-5)[well].all()
        assert np.isfinite(g).all() and (np.abs(np.linalg.norm(g[:, :3], axis=1) - 1) < 1e-5).all()
        # curvature is a ratio of eigenvalues: defined for every k >= 2
        assert np.allclose(g[sel, 3], r[:, 3], atol=5e-5)


def test_ties_at_the_kth_distance_pick_the_lowest_indices(capi, orc):
    """A lattice: many neighbours at exactly the same distance.  The plane of k = 6 of the 8 kept candidates changes
    with the choice among the tied ones, so the fused cut must be the (d2, index) order of the search."""
    g1 = np.arange(24, dtype=np.float32)
    pts = np.stack(np.meshgrid(g1, g1, g1[:6], indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32)
    pts = pts[np.random.default_rng(1).permutation(len(pts))]  # index order unrelated to position
    t = capi.Tree(pts)
    vp = (12.0, 12.0, 100.0)
    for k in (6, 5, 12):
        g = capi.normals(t, pts, k=k, viewpoint=vp)
        sel = np.random.default_rng(k).choice(len(pts), 1200, replace=False)
        lists = knn_lists(orc, pts, pts[sel], k)
        r = numpy_normals(pts, pts[sel], lists, vp)
        # curvature (eigenvalue ratio) is unique even where the eigenvector is not (symmetric neighbourhoods)
        assert np.allclose(g[sel, 3], r[:, 3], atol=5e-5), np.abs(g[sel, 3] - r[:, 3]).max()


@pytest.mark.parametrize("radius", [0.6, 1.5])
def test_radius_normals_dense_and_sparse(capi, orc, radius):
    pts = wavy(120_000, 9)  # ~33 points per unit area: 37 / 235 neighbours - the larger drains the buffer repeatedly
    vp = (30.0, 30.0, 500.0)
    t = capi.Tree(pts)
    rng = np.random.default_rng(2)
    for q in (pts, pts[rng.choice(len(pts), 5_000, replace=False)]):
        q = q.copy()
        q[11] = np.nan
        g = capi.normals(t, q, radius=radius, viewpoint=vp)
        assert np.isnan(g[11]).all()
        sel = rng.choice(len(q), 1200, replace=False)
        sel = sel[sel != 11]
        lists = radius_lists(orc, pts, q[sel], radius)
        r = numpy_normals(pts, q[sel], lists, vp)
        enough = np.array([len(l) >= 8 for l in lists])
        assert enough.mean() > 0.95
        assert close_up_to_view_sign(g[sel], r, q[sel], vp, 5e-5)[enough].all()
    # a query with no neighbour inside the radius: NaN row, as normal_3d.hpp:62-68
    far = np.array([[1e3, 1e3, 1e3], pts[0]], np.float32)
    gf = capi.normals(t, far, radius=radius, viewpoint=vp)
    assert np.isnan(gf[0]).all() and np.isfinite(gf[1]).all()
