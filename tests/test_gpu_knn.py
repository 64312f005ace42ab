"""GPU parity: pclmi_knn (HIP, through the C ABI) vs the CPU oracle, bit-exact."""
import json

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


def _check(capi, orc, pts, q, k, brute=True):
    t = capi.Tree(pts)
    gi, gd = t.knn(q, k)
    oi, od = (orc.knn_brute if brute else (lambda a, b, kk: orc.Tree(a).knn(b, kk, nthreads=8)))(pts, q, k)
    assert gi.shape == oi.shape
    bad = np.nonzero((gi != oi).any(axis=1) | (gd.view(np.int32) != od.view(np.int32)).any(axis=1))[0]
    assert len(bad) == 0, "first mismatch at query %d:\n gpu %s %s\n cpu %s %s" % (
        bad[0], gi[bad[0]], gd[bad[0]], oi[bad[0]], od[bad[0]])
    t.close()


def test_known_answer_10_points(capi):
    kat = util.golden("kdtree_kat.json")
    pts = np.array(kat["points"], np.float32)
    q = np.array([kat["query"]], np.float32)
    for case in kat["cases"]:
        alpha = np.array(case["alpha"], np.float32)
        p, qq = pts * alpha, q * alpha
        if case["dims"] == 2:
            p = p.copy(); p[:, 2] = 0
            qq = qq.copy(); qq[:, 2] = 0
        t = capi.Tree(p)
        idx, d2 = t.knn(qq, kat["k"])
        assert idx[0].tolist() == case["indices"], case["name"]
        assert np.allclose(d2[0], case["distances"], atol=kat["distance_tol"])


def test_bun_1nn_matches_golden(capi):
    api = util.golden("registration_api.json")
    src, tgt = util.bun0(), util.bun4()
    t = capi.Tree(tgt)
    idx, d2 = t.knn(src, 1)
    exp = np.array(api["correspondences_original"])
    assert idx[:, 0].tolist() == exp[:, 1].tolist()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 10, 16, 20, 32, 33, 64, 512])
def test_uniform_1200_all_k(capi, orc, k):
    # recipe of test/test_search.cpp:571-617 (1200 uniform points), k list of :624-631 and more
    pts = util.uniform_cloud(1200, 11)
    q = util.uniform_cloud(777, 12)
    _check(capi, orc, pts, q, k)


@pytest.mark.parametrize("k", [1, 8, 16, 20, 32, 40])
def test_grid_ties(capi, orc, k):
    # 11^3 grid: every query has many exactly-equal distances -> exercises (d2, index) order
    pts = util.grid_cloud()
    q = np.concatenate([pts[::7], np.array([[0.01, 0.01, 0.01], [0, 0, 0], [0.05, 0.05, 0.05]], np.float32),
                        pts[:200] + np.float32(0.05)])
    _check(capi, orc, pts, q, k)


@pytest.mark.parametrize("k", [2, 3, 4, 5, 8, 10, 16, 20, 32])
def test_sparse_batches_take_the_per_lane_kernel(capi, orc, k):
    """Fewer than a sixth as many queries as target points.  More than 4096 of them: knn_lane_ix_kernel (each lane
    walks for itself); fewer: the union kernel with the batch spread over the wavefronts, down to one query each.  The
    same bars as the dense union kernel: ties on a grid, duplicates beyond a bucket, NaN queries, a ragged last wave -
    and the kernels must agree on one batch (PCLMI_KNN is read once per process, so the batch sizes pick the kernels)."""
    pts = util.grid_cloud()  # 1331 points, many exactly-equal distances
    q = np.concatenate([pts[::13], np.array([[0.01, 0.01, 0.01], [0, 0, 0], [0.05, 0.05, 0.05]], np.float32),
                        pts[:100] + np.float32(0.05)])
    assert len(q) * 6 < len(pts)
    _check(capi, orc, pts, q, k)
    rng = np.random.default_rng(61)
    base = rng.random((60, 3), dtype=np.float32)
    dup = np.repeat(base, 40, axis=0)  # 2400 points, each value 40 times
    qq = np.concatenate([base, rng.random((139, 3), dtype=np.float32)])
    qq[17] = np.nan
    t = capi.Tree(dup)
    gi, gd = t.knn(qq, k)
    good = np.isfinite(qq).all(axis=1)
    oi, od = orc.knn_brute(dup, qq[good], k)
    assert np.array_equal(gi[good], oi) and np.array_equal(gd[good].view(np.int32), od.view(np.int32))
    assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all()
    big = util.uniform_cloud(120_000, 62, 0, 100)
    tb = capi.Tree(big)
    sparse = util.uniform_cloud(15_001, 63, 0, 100)   # 15 001 * 6 < 120 000: per-lane kernel
    dense = np.concatenate([sparse, util.uniform_cloud(20_000, 64, 0, 100)])  # 35 001 * 6 > 120 000: union kernel
    si, sd = tb.knn(sparse, k)
    di, dd = tb.knn(dense, k)
    assert np.array_equal(si, di[:15_001]) and np.array_equal(sd.view(np.int32), dd[:15_001].view(np.int32))


@pytest.mark.parametrize("k", [1, 16, 40, 64])
def test_small_sparse_batches_are_spread_over_the_wavefronts(capi, orc, k):
    """Sparse batches of 1 .. 4096 queries run with 1, 2, 4 ... queries per wavefront (lane_queries_per_wave), above
    that the per-lane kernels take over (k <= 32) or the union kernel fills its wavefronts again (k = 33..64): every
    batch size around those switches must give the oracle's rows, NaN queries included."""
    pts = util.uniform_cloud(120_000, 71, 0, 100)
    t = capi.Tree(pts)
    ot = orc.Tree(pts)
    allq = util.uniform_cloud(9000, 72, -5, 105)  # some outside the target's box
    allq[5] = np.nan
    allq[4090, 1] = np.inf
    for nq in (1, 2, 3, 5, 63, 64, 65, 1000, 2049, 4095, 4096, 4097, 9000):
        q = allq[:nq]
        gi, gd = t.knn(q, k)
        good = np.isfinite(q).all(axis=1)
        oi, od = ot.knn(q[good], k, nthreads=8)
        assert np.array_equal(gi[good], oi) and np.array_equal(gd[good].view(np.int32), od.view(np.int32)), (k, nq)
        assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all(), (k, nq)
    t.close()


def test_duplicates_and_tiny_clouds(capi, orc):
    rng = np.random.default_rng(5)
    base = rng.random((40, 3), dtype=np.float32)
    pts = np.repeat(base, 30, axis=0)  # 1200 points, each value 30 times: ties beyond any bucket
    q = np.concatenate([base, rng.random((100, 3), dtype=np.float32)])
    for k in (1, 4, 16, 31, 32, 45):
        _check(capi, orc, pts, q, k)
    for n in (1, 2, 3, 15, 16, 17, 33):
        p = rng.random((n, 3), dtype=np.float32)
        for k in (1, 2, 16, 50):
            _check(capi, orc, p, q[:70], k)


def test_nan_rows_and_index_subset(capi, orc):
    # sparse cloud of test/test_search.cpp:582-599: every 11th point NaN; plus an index view
    pts = util.uniform_cloud(1200, 21)
    pts[::11] = np.nan
    q = util.uniform_cloud(300, 22)
    t = capi.Tree(pts)
    valid = np.nonzero(np.isfinite(pts).all(axis=1))[0]
    assert len(t) == len(valid)
    gi, gd = t.knn(q, 8)
    oi, od = orc.knn_brute(pts[valid], q, 8)
    assert np.array_equal(gi, valid[oi]) and np.array_equal(gd, od)
    rng = np.random.default_rng(3)
    view = rng.permutation(1200)[:120].astype(np.int32)
    t2 = capi.Tree(pts, indices=view)
    vv = np.array([i for i in view if np.isfinite(pts[i]).all()])
    gi, gd = t2.knn(q, 5)
    assert np.isin(gi, vv).all()
    # distances are exact; indices are rows of the whole cloud
    oi, od = orc.knn_brute(pts[vv], q, 5)
    assert np.array_equal(gd, od)
    recomputed = ((q[:, None, :] - pts[gi]) ** 2)
    assert np.array_equal(((recomputed[..., 0] + recomputed[..., 1]) + recomputed[..., 2]).astype(np.float32), gd)


def test_empty_and_errors(capi):
    t = capi.Tree(np.zeros((0, 3), np.float32))
    assert len(t) == 0
    idx, d2 = t.knn(np.zeros((4, 3), np.float32), 3)
    assert idx.shape == (4, 0)
    t2 = capi.Tree(util.uniform_cloud(100, 1))
    with pytest.raises(capi.PclmiError):
        t2.knn(np.zeros((4, 3), np.float32), 0)
    idx, d2 = t2.knn(np.zeros((0, 3), np.float32), 3)
    assert idx.shape == (0, 3)
    idx, d2 = t2.knn(util.uniform_cloud(5, 2), 1000)  # k clamped to 100 (kdtree_flann.hpp:85-86)
    assert idx.shape == (5, 100) and np.all(np.sort(idx, axis=1) == np.arange(100))


@pytest.mark.parametrize("n,nq,k", [(100_000, 20_000, 16), (100_000, 20_000, 1), (250_000, 10_000, 32)])
def test_large_vs_oracle_tree(capi, orc, n, nq, k):
    # uniform cube [0,1024)^3 as BASELINE C2 (scaled down so the CPU oracle finishes in seconds)
    pts = util.uniform_cloud(n, 20260101, 0, 1024)
    q = util.uniform_cloud(nq, 20260102, 0, 1024)
    _check(capi, orc, pts, q, k, brute=False)


def test_large_skewed_batches_through_the_two_level_query_order(capi, orc):
    """Batches of >= 32768 queries take the two-level counting sort (knn_kernels.hip tile_count_kernel ...): every
    query must come out exactly once whatever the distribution - all queries in one tiny corner (one high bin),
    queries far outside the tree's box, non-finite rows, a batch size that is no multiple of the tile."""
    pts = util.uniform_cloud(200_000, 51, 0, 1024)
    tree, otree = capi.Tree(pts), orc.Tree(pts)
    rng = np.random.default_rng(52)
    corner = (rng.random((40_001, 3), dtype=np.float32) * 3 + 500).astype(np.float32)        # one cell or two
    spread = (rng.random((50_000, 3), dtype=np.float32) * 3000 - 1000).astype(np.float32)     # mostly outside the box
    mixed = np.concatenate([corner[:20_000], spread[:20_000], util.uniform_cloud(30_003, 53, 0, 1024)])
    mixed[7] = np.nan
    mixed[33_000, 2] = np.inf
    for q in (corner, spread, mixed):
        for k in (16, 1):
            gi, gd = tree.knn(q, k)
            good = np.isfinite(q).all(axis=1)
            sel = np.flatnonzero(good)[:: max(1, good.sum() // 4000)]  # the oracle checks a sample ...
            oi, od = otree.knn(q[sel], k, nthreads=8)
            assert np.array_equal(gi[sel], oi) and np.array_equal(gd[sel].view(np.int32), od.view(np.int32))
            assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all()
            # ... and every finite row was written: first neighbour recomputed from the coordinates
            d = q[good] - pts[gi[good][:, 0]]
            rec = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
            assert np.array_equal(rec, gd[good][:, 0])
    # exactly the threshold size; a tree whose box has no extent (every cell number is 0: one bin holds the batch)
    q = util.uniform_cloud(32_768, 54, 0, 1024)
    gi, gd = tree.knn(q, 8)
    oi, od = otree.knn(q[:3000], 8, nthreads=8)
    assert np.array_equal(gi[:3000], oi) and np.array_equal(gd[:3000].view(np.int32), od.view(np.int32))
    flat = capi.Tree(np.full((50, 3), 7.0, np.float32))
    gi, gd = flat.knn(q, 3)
    assert (gi == np.array([0, 1, 2])).all() and np.array_equal(gd[:, 0], gd[:, 2])
    # radius search shares the ordering
    ro, ri, rd = tree.radius(corner, 12.0)
    oo, oi, od = otree.radius(corner[:2000], 12.0)
    assert np.array_equal(ro[:2001], oo) and np.array_equal(ri[:oo[-1]], oi) and np.array_equal(rd[:oo[-1]], od)


def test_surface_like_and_far_queries(capi, orc):
    rng = np.random.default_rng(9)
    # points on a few planes + noise (real clouds are 2-manifolds), queries partly far outside
    n = 60_000
    uv = rng.random((n, 2), dtype=np.float32) * 100
    plane = rng.integers(0, 4, n)
    pts = np.zeros((n, 3), np.float32)
    pts[:, 0] = uv[:, 0]
    pts[:, 1] = uv[:, 1]
    pts[:, 2] = plane * 7 + rng.normal(0, 0.05, n).astype(np.float32)
    q = np.concatenate([pts[::50] + np.float32(0.3), rng.random((500, 3), dtype=np.float32) * 1000 - 300])
    _check(capi, orc, pts, q.astype(np.float32), 16, brute=False)
    _check(capi, orc, pts, q.astype(np.float32), 1, brute=False)


def test_partial_wave_and_nan_queries_do_not_walk_the_whole_tree(capi, orc):
    # regression: lanes past the end / NaN queries once "needed" every box (box distance of NaN is 0)
    import time
    pts = util.uniform_cloud(300_000, 31, 0, 1024)
    q = util.uniform_cloud(10_047, 32, 0, 1024)  # not a multiple of 64
    q[5] = np.nan
    q[4000, 1] = np.inf
    t = capi.Tree(pts)
    for k in (1, 16):
        t.knn(q, k)  # warm up
        t0 = time.time()
        gi, gd = t.knn(q, k)
        dt = time.time() - t0
        good = np.isfinite(q).all(axis=1)
        oi, od = orc.Tree(pts).knn(q[good], k, nthreads=8)
        assert np.array_equal(gi[good], oi) and np.array_equal(gd[good], od)
        assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all()
        assert dt < 0.05, "k=%d took %.1f ms" % (k, dt * 1e3)


def test_concurrent_const_queries_on_one_tree(capi, orc):
    """The reference's const search methods are called concurrently from OpenMP loops on one tree
    (features/.../impl/normal_3d_omp.hpp:62-68, test/test_search.cpp:308-325); every call taking a
    const pclmi_tree* must be re-entrant.  ctypes releases the GIL, so these threads really overlap."""
    import threading
    pts = util.uniform_cloud(200_000, 77, 0, 100)
    t = capi.Tree(pts)
    qs = [util.uniform_cloud(3000 + 137 * i, 100 + i, 0, 100) for i in range(8)]
    serial = [(t.knn(q, 16), t.knn(q, 1), t.radius(q, 1.5)) for q in qs]
    got, errs = [None] * len(qs), []

    def work(i):
        try:
            for _ in range(3):
                got[i] = (t.knn(qs[i], 16), t.knn(qs[i], 1), t.radius(qs[i], 1.5))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(qs))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for s, g in zip(serial, got):
        for a, b in zip(s, g):
            for u, v in zip(a, b):
                assert np.array_equal(u, v, equal_nan=True)
    oi, od = orc.Tree(pts).knn(qs[0], 16, nthreads=8)
    assert np.array_equal(got[0][0][0], oi)


@pytest.mark.parametrize("k", [65, 128, 200, 512])
def test_k_above_64_in_tiles_of_64(capi, orc, k):
    """k > 64 on batches of >= 16 384 queries: the list in tiles of 64, every tile after the first restricted to the
    (d2, index) keys above the last key of the tile before it (launch_union_tiled).  Exact ties ACROSS a tile boundary
    are the point: an 11^3 lattice (dozens of equal distances per query), a cloud of 40-fold duplicates (equal
    distances, ascending indices), NaN queries, k that is not a multiple of 64 - bit-exact against brute force; and the
    same queries as a small batch (the heap kernel) must give the same rows."""
    pts = util.grid_cloud()  # 1331 points
    rng = np.random.default_rng(70 + k)
    q = np.concatenate([pts, pts[:669] + np.float32(0.05), rng.random((15_000, 3), dtype=np.float32) - np.float32(0.5)])
    q[5] = np.nan
    assert len(q) >= 16_384
    t = capi.Tree(pts)
    gi, gd = t.knn(q, k)
    good = np.isfinite(q).all(axis=1)
    sel = np.flatnonzero(good)[::3]
    oi, od = orc.knn_brute(pts, q[sel], k)
    assert np.array_equal(gi[sel], oi) and np.array_equal(gd[sel].view(np.int32), od.view(np.int32))
    assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all()
    hi, hd = t.knn(q[:2000], k)  # small batch: the per-lane heap kernel
    ok = good[:2000]
    assert np.array_equal(hi[ok], gi[:2000][ok]) and np.array_equal(hd[ok].view(np.int32), gd[:2000][ok].view(np.int32))
    base = rng.random((60, 3), dtype=np.float32)
    dup = np.repeat(base, 40, axis=0)  # 2400 points, each value 40 times: ties of 40 straddle the tile boundaries
    qq = np.concatenate([base, rng.random((16_400, 3), dtype=np.float32)])
    t2 = capi.Tree(dup)
    gi, gd = t2.knn(qq, k)
    oi, od = orc.knn_brute(dup, qq[::5], k)
    assert np.array_equal(gi[::5], oi) and np.array_equal(gd[::5].view(np.int32), od.view(np.int32))
    big = util.uniform_cloud(200_000, 71, 0, 100)
    tb = capi.Tree(big)
    qb = util.uniform_cloud(40_000, 72, 0, 100)
    gi, gd = tb.knn(qb, k)
    oi, od = orc.Tree(big).knn(qb[:3000], k, nthreads=16)
    assert np.array_equal(gi[:3000], oi) and np.array_equal(gd[:3000].view(np.int32), od.view(np.int32))


def test_queries_that_are_no_points_at_k_above_64(capi, orc):
    """Every k-NN kernel answers a non-finite query with (-1, NaN) rows; the heap kernel (k > 64, small batches) once
    returned real indices with NaN distances (found by tools/stress_search.py)."""
    pts = util.uniform_cloud(700, 61)
    tree = capi.Tree(pts)
    q = util.uniform_cloud(40, 62)
    q[0] = np.nan
    q[17, 1] = np.inf
    good = np.isfinite(q).all(axis=1)
    for k in (65, 200, 700):
        gi, gd = tree.knn(q, k)
        oi, od = orc.knn_brute(pts, q[good], k)
        assert np.array_equal(gi[good], oi) and np.array_equal(gd[good].view(np.int32), od.view(np.int32))
        assert (gi[~good] == -1).all() and np.isnan(gd[~good]).all()
    gi, gd = tree.knn(q[:1], 65)  # a batch of one, and that one no point
    assert (gi == -1).all() and np.isnan(gd).all()
