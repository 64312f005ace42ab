"""The device tree builder (tree_build_dev.hip, pclmi_tree_create / pclmi_tree_create_dev) against the host builder
(tree_build.cpp, kept behind PCLMI_TREE_BUILD=host): the SAME tree - every node box and every leaf record bit for
bit - for clouds with ties, duplicate points, non-finite rows, index views and sizes around the bucket and level
boundaries; and, as search results must never depend on the tree, bit-exact k-NN / radius results against the oracle
on the device-built tree.  Replaces the FLANN index build of pcl::KdTreeFLANN::setInputCloud
(kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:46-75,180-249).
"""
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from pcl_amd import capi as c
    assert c.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return c


def build_both(capi, pts, indices=None):
    os.environ.pop("PCLMI_TREE_BUILD", None)
    dev = capi.Tree(pts, indices)
    os.environ["PCLMI_TREE_BUILD"] = "host"
    try:
        host = capi.Tree(pts, indices)
    finally:
        os.environ.pop("PCLMI_TREE_BUILD", None)
    return dev, host


def assert_same_tree(dev, host):
    assert len(dev) == len(host)
    bd, ld = dev.debug_arrays()
    bh, lh = host.debug_arrays()
    assert bd.shape == bh.shape and ld.shape == lh.shape
    # boxes by VALUE: where a box edge is zero the two builders may keep zeros of different sign (min / max of a mix of
    # -0 and +0), which no distance computation can tell apart
    assert np.array_equal(bd[1:], bh[1:]), "node boxes differ"
    assert np.array_equal(bd[0], bd[1])  # id 0 mirrors the root
    assert np.array_equal(ld.view(np.int32), lh.view(np.int32)), "leaf records differ"


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 31, 33, 100, 255, 256, 257, 1000, 4097, 65_536, 100_003, 1_000_000])
def test_device_tree_equals_host_tree(capi, n):
    pts = util.uniform_cloud(n, n % 97 + 1, -50, 50)
    dev, host = build_both(capi, pts)
    assert_same_tree(dev, host)


def test_ties_duplicates_signed_zeros_and_nonfinite_rows(capi):
    rng = np.random.default_rng(3)
    # a lattice (every coordinate value shared by thousands of points) with every point stored twice
    g = np.arange(20, dtype=np.float32) - 10
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    pts = np.concatenate([lat, lat, rng.normal(0, 3, (5000, 3)).astype(np.float32)])
    pts = pts[rng.permutation(len(pts))].copy()
    pts[pts == 0] = np.where(rng.random((pts == 0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
    pts[::97, 1] = np.nan
    pts[5::131, 0] = np.inf
    pts[7::211, 2] = -np.inf
    dev, host = build_both(capi, pts)
    assert len(dev) == np.isfinite(pts).all(axis=1).sum()
    assert_same_tree(dev, host)
    # an ascending index view (the IndicesConstPtr overload, kdtree_flann.hpp:214-249)
    view = np.flatnonzero(rng.random(len(pts)) < 0.4).astype(np.int32)
    dev, host = build_both(capi, pts, view)
    assert_same_tree(dev, host)
    # an index outside the cloud is refused
    with pytest.raises(capi.PclmiError):
        capi.Tree(pts, np.array([0, len(pts)], np.int32))
    # all rows non-finite / no rows: a valid empty tree
    assert len(capi.Tree(np.full((10, 3), np.nan, np.float32))) == 0
    assert len(capi.Tree(np.zeros((0, 3), np.float32))) == 0


def test_unordered_index_view_and_searches_against_the_oracle(capi):
    """With an index list that is not ascending (or repeats rows) the in-bucket order of the two builders may differ;
    search results may not."""
    from oracle import pcl_oracle as orc
    rng = np.random.default_rng(8)
    pts = util.uniform_cloud(50_000, 4, 0, 100)
    view = rng.permutation(len(pts))[:30_000].astype(np.int32)
    t = capi.Tree(pts, view)
    assert len(t) == len(view)
    q = (rng.random((4000, 3), dtype=np.float32) * 100).astype(np.float32)
    gi, gd = t.knn(q, 16)
    ot = orc.Tree(pts[view])  # the oracle indexes the selected rows; its answers are positions in the view
    oi, od = ot.knn(q, 16, nthreads=8)
    assert np.array_equal(gi, view[oi]) and np.array_equal(gd.view(np.int32), od.view(np.int32))
    off, ri, rd = t.radius(q[:1000], 3.0)
    ooff, ori, ord_ = ot.radius(q[:1000], 3.0, nthreads=8)
    assert np.array_equal(off, ooff) and np.array_equal(rd.view(np.int32), ord_.view(np.int32))
    assert np.array_equal(ri, view[ori])
    # a view that repeats rows indexes them twice (the reference copies whatever the list names)
    twice = np.concatenate([view[:100], view[:100]]).astype(np.int32)
    t2 = capi.Tree(pts, twice)
    assert len(t2) == 200
    i2, d2 = t2.knn(pts[view[:100]], 2)
    assert np.array_equal(i2[:, 0], view[:100]) and np.array_equal(i2[:, 1], view[:100]) and (d2 == 0).all()


def test_tree_from_device_memory(capi):
    import torch
    from oracle import pcl_oracle as orc
    pts = util.uniform_cloud(200_000, 21, 0, 100)
    pts[::1000, 0] = np.nan
    d = torch.from_numpy(np.concatenate([pts, np.ones((len(pts), 1), np.float32)], axis=1)).cuda()  # 16-byte rows
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d2 = d.clone()  # produced on the stream the build is ordered on
        t = capi.Tree.from_device(d2.data_ptr(), len(pts), 16, stream=s.cuda_stream)
    host = capi.Tree(pts)
    assert_same_tree(t, host)
    q = pts[5:3005]
    gi, gd = t.knn(np.nan_to_num(q), 8)
    fin = np.flatnonzero(np.isfinite(pts).all(axis=1))  # the oracle indexes the finite rows (an ascending view)
    oi, od = orc.Tree(pts[fin]).knn(np.nan_to_num(q), 8, nthreads=8)
    assert np.array_equal(gi, fin[oi]) and np.array_equal(gd.view(np.int32), od.view(np.int32))
    # device index list
    view = torch.arange(0, len(pts), 3, dtype=torch.int32, device="cuda")
    tv = capi.Tree.from_device(d.data_ptr(), len(pts), 16, view.data_ptr(), len(view))
    hv = capi.Tree(pts, view.cpu().numpy())
    assert_same_tree(tv, hv)
