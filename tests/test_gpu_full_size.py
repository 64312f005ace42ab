"""GPU parity at BASELINE.json's FULL sizes (configs[1..4]; SURVEY 8d C2-C5, C5 as one query batch).

The CPU oracle cannot check 16M..64M results in seconds, so every full-size output is checked
 (a) exhaustively through size-independent properties - (d2, index) order, index range and uniqueness,
     every distance recomputed bit-exactly from the coordinates, CSR shape, strict radius,
     the radius-search prefix equal to the kNN list of the same query (two different kernels),
     invariance under a permutation of the queries - and
 (b) bit-exactly against the oracle's FLANN-style tree on a seeded sample of the queries.
Inputs are the bench's own (seeds 20260101 / 20260102, uniform in [0,1024)^3).
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SEED_T, SEED_Q = 20260101, 20260102


@pytest.fixture(scope="module")
def capi():
    from pcl_amd import capi as c
    assert c.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return c


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


@pytest.fixture(scope="module")
def c2(capi):
    pts = util.uniform_cloud(1_000_000, SEED_T, 0, 1024)
    q = util.uniform_cloud(1_000_000, SEED_Q, 0, 1024)
    return pts, q, capi.Tree(pts)


def recompute_d2(q, p):
    """((dx*dx + dy*dy) + dz*dz) in fp32, one rounding per operation (flann::L2_Simple<float>)."""
    d = (q - p).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def check_knn_properties(pts, q, idx, d2, k):
    n = len(pts)
    assert idx.shape == (len(q), k) and d2.shape == (len(q), k)
    assert idx.min() >= 0 and idx.max() < n
    # ascending in the total order (d2, index)
    dd = np.diff(d2, axis=1)
    assert (dd >= 0).all()
    ties = dd == 0
    assert (np.diff(idx, axis=1)[ties] > 0).all(), "equal distances must come in ascending index order"
    # no neighbour twice
    assert (np.diff(np.sort(idx, axis=1), axis=1) > 0).all()
    # every distance is the fp32 distance of the point it names
    step = 250_000
    for a in range(0, len(q), step):
        rec = recompute_d2(q[a:a + step, None, :], pts[idx[a:a + step]])
        assert np.array_equal(rec.view(np.int32), d2[a:a + step].view(np.int32))


def test_c2_knn16_1m_x_1m(capi, orc, c2):
    pts, q, tree = c2
    k = 16
    idx, d2 = tree.knn(q, k)
    check_knn_properties(pts, q, idx, d2, k)
    # seeded sample against the oracle's FLANN-style tree: bit-exact indices and distances
    sel = np.random.default_rng(5).choice(len(q), 20_000, replace=False)
    oi, od = orc.Tree(pts).knn(q[sel], k, nthreads=16)
    assert np.array_equal(idx[sel], oi) and np.array_equal(d2[sel].view(np.int32), od.view(np.int32))
    # results do not depend on the order (or the batch) the queries arrive in
    perm = np.random.default_rng(6).permutation(len(q))[:300_000]
    pi, pd = tree.knn(q[perm], k)
    assert np.array_equal(pi, idx[perm]) and np.array_equal(pd.view(np.int32), d2[perm].view(np.int32))
    # k = 1 (the ICP correspondence search) is the first column
    i1, d1 = tree.knn(q, 1)
    assert np.array_equal(i1[:, 0], idx[:, 0]) and np.array_equal(d1[:, 0].view(np.int32), d2[:, 0].view(np.int32))


def test_c3_radius_1m_x_1m_about_32_neighbours(capi, orc, c2):
    pts, q, tree = c2
    r = float((32 * 3 / (4 * np.pi * len(pts) / 1024.0 ** 3)) ** (1 / 3))  # E[m] = 32
    r2 = np.float32(r * r)
    off, idx, d2 = tree.radius(q, r)
    nq = len(q)
    assert off.shape == (nq + 1,) and off[0] == 0 and off[-1] == len(idx) == len(d2)
    cnt = np.diff(off.astype(np.int64))
    assert (cnt >= 0).all() and 28 < cnt.mean() < 34
    assert idx.min() >= 0 and idx.max() < len(pts)
    assert (d2 < r2).all(), "strict radius (kdtree_flann.hpp:140)"
    seg = np.repeat(np.arange(nq), cnt)
    # ascending (d2, index) inside every segment
    same = seg[1:] == seg[:-1]
    dd = np.diff(d2)
    assert (dd[same] >= 0).all()
    assert (np.diff(idx)[same & (dd == 0)] > 0).all()
    # every distance recomputed
    step = 4_000_000
    for a in range(0, len(idx), step):
        rec = recompute_d2(q[seg[a:a + step]], pts[idx[a:a + step]])
        assert np.array_equal(rec.view(np.int32), d2[a:a + step].view(np.int32))
    # two kernels, one answer: the first min(count, 16) radius results are the kNN list of that query,
    # and a query with fewer than 16 results has its 16th neighbour outside the radius
    ki, kd = tree.knn(q, 16)
    first = off[:-1].astype(np.int64)
    for j in range(16):
        has = cnt > j
        assert np.array_equal(idx[first[has] + j], ki[has, j])
        assert np.array_equal(d2[first[has] + j].view(np.int32), kd[has, j].view(np.int32))
        assert (kd[~has, j] >= r2).all()
    # seeded sample against the oracle tree, bit-exact CSR
    sel = np.random.default_rng(7).choice(nq, 5_000, replace=False)
    oo, oi, od = orc.Tree(pts).radius(q[sel], r, nthreads=16)
    assert np.array_equal(np.diff(oo.astype(np.int64)), cnt[sel])
    take = np.concatenate([np.arange(first[s], first[s] + cnt[s]) for s in sel])
    assert np.array_equal(idx[take], oi) and np.array_equal(d2[take].view(np.int32), od.view(np.int32))
    # max_nn and unsorted variants on a slice: the nearest 8, and ascending index
    sub = q[:50_000]
    o8, i8, d8 = tree.radius(sub, r, 8, True)
    c8 = np.diff(o8.astype(np.int64))
    assert np.array_equal(c8, np.minimum(cnt[:50_000], 8))
    f8 = o8[:-1].astype(np.int64)
    for j in range(8):
        has = c8 > j
        assert np.array_equal(i8[f8[has] + j], ki[:50_000][has, j])
    ou, iu, du = tree.radius(sub, r, 0, False)
    assert np.array_equal(ou, off[:50_001])
    segu = np.repeat(np.arange(50_000), cnt[:50_000])
    assert (np.diff(iu)[segu[1:] == segu[:-1]] > 0).all()
    assert np.array_equal(np.sort(iu + segu * len(pts)), np.sort(idx[:len(iu)] + seg[:len(iu)] * len(pts)))


def test_c5_knn32_10m_target_one_query_batch(capi, orc):
    # configs[4]: 10M-point target, k = 32; the 100M queries arrive as batches per GPU - one 2M batch here
    pts = util.uniform_cloud(10_000_000, SEED_T, 0, 2048)
    q = util.uniform_cloud(2_000_000, SEED_Q, 0, 2048)
    tree = capi.Tree(pts)
    k = 32
    idx, d2 = tree.knn(q, k)
    check_knn_properties(pts, q, idx, d2, k)
    sel = np.random.default_rng(8).choice(len(q), 4_000, replace=False)
    oi, od = orc.Tree(pts).knn(q[sel], k, nthreads=16)
    assert np.array_equal(idx[sel], oi) and np.array_equal(d2[sel].view(np.int32), od.view(np.int32))
    tree.close()


def test_c4_icp_1m_x_1m_30_iterations(capi, orc):
    # configs[3] on one GPU and as two shards: bench.py's clouds (surface-like target, source = target moved
    # by a known rigid motion + noise), 30 iterations forced, RANSAC off
    import bench
    n = 1_000_000
    tgt = bench.surface_cloud(n, SEED_T + 10)
    T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
    c = np.array([512.0, 512.0, 512.0])
    T0[:3, 3] += c - T0[:3, :3] @ c
    rng = np.random.default_rng(SEED_Q + 10)
    src = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
    Ti = np.linalg.inv(T0)
    src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    tree = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 30, 25.0, 0
    rc, T, out, st = capi.icp_align(tree, src, p)
    assert rc == 0 and st.converged and st.iterations == 30
    assert np.abs(T - T0).max() < 1e-3, "the known motion is recovered (noise-limited)"
    assert T[3].tolist() == [0, 0, 0, 1]
    # against the oracle's restatement of the reference loop (Scalar = double estimator): 1e-5
    po = orc.icp_default_params()
    po.max_iterations, po.max_correspondence_distance, po.ransac_iterations, po.svd_scalar_double = 30, 25.0, 0, 1
    rco, To, outo, sto = orc.icp_align(src, tgt, po, nthreads=16)
    assert rco == 0 and sto.iterations == 30
    assert st.last_correspondences == sto.last_correspondences
    assert np.abs(T - To).max() <= 1e-5, np.abs(T - To).max()
    assert np.abs(out[:, :3] - outo[:, :3]).max() <= 1e-3  # fp32 points of magnitude 1e3: a few ulps
    # the aligned cloud is the source moved by the accumulated transform chain: its 1-NN distances to the
    # target equal a fresh search
    i1, d1 = tree.knn(out[:, :3], 1)
    assert np.array_equal(d1[:, 0].view(np.int32), recompute_d2(out[:, :3], tgt[i1[:, 0]]).view(np.int32))
    # two shards + summed moments (what two ranks compute) give the same transform
    half = n // 2
    a, b = capi.IcpShard(tree, src[:half]), capi.IcpShard(tree, src[half:])
    Tf = np.eye(4, dtype=np.float32)
    from pcl_amd import sharded
    for _ in range(30):
        s = a.step(25.0) + b.step(25.0)
        Ti_ = capi.icp_solve(s)
        a.set_transform(Ti_)
        b.set_transform(Ti_)
        Tf = sharded.mat4_mul_f32(Ti_, Tf)
    a.end(want_aligned=False)
    b.end(want_aligned=False)
    assert np.abs(Tf - T).max() <= 1e-5


def c4_clouds():
    import bench
    n = 1_000_000
    tgt = bench.surface_cloud(n, SEED_T + 10)
    T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
    c = np.array([512.0, 512.0, 512.0])
    T0[:3, 3] += c - T0[:3, :3] @ c
    rng = np.random.default_rng(SEED_Q + 10)
    src = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
    Ti = np.linalg.inv(T0)
    src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return tgt, src, T0


def test_c4_icp_1m_x_1m_ransac_on(capi, orc):
    # configs[3]'s clouds with the reference's DEFAULT rejection step (icp.h:104: 1000 trials, icp.hpp:113-150):
    # the first iterations need many trials (few inliers: host-fed batches), the later ones settle with the first
    # model inside the fused kernel.  Against the oracle's one-trial-at-a-time loop: same pairs, T within 1e-5.
    tgt, src, T0 = c4_clouds()
    tree = capi.Tree(tgt)
    p, po = capi.icp_default_params(), orc.icp_default_params()
    for x in (p, po):
        x.max_iterations, x.max_correspondence_distance, x.ransac_iterations = 5, 25.0, 1000
        x.ransac_outlier_threshold = 0.5
    po.svd_scalar_double = 1
    rc, T, out, st = capi.icp_align(tree, src, p)
    rco, To, outo, sto = orc.icp_align(src, tgt, po, nthreads=16)
    assert rc == 0 and rco == 0 and st.iterations == sto.iterations == 5
    assert st.last_correspondences == sto.last_correspondences
    assert 3 <= st.last_inliers <= st.last_correspondences
    assert np.abs(T - To).max() <= 1e-5, np.abs(T - To).max()
    assert np.abs(T - T0).max() < 5e-3, "five iterations with the rejection step are close to the known motion"
    tree.close()


def test_c4_icp_1m_x_1m_point_to_plane(capi, orc):
    # configs[3]'s clouds with TransformationEstimationPointToPlaneLLS in the loop (registration.h:140-160), target
    # normals from the fused k = 16 normal estimation; 10 iterations against the oracle's loop
    tgt, src, T0 = c4_clouds()
    tree = capi.Tree(tgt)
    tn = capi.normals(tree, tgt, k=16, viewpoint=(512.0, 512.0, 5000.0))[:, :3].copy()
    p, po = capi.icp_default_params(), orc.icp_default_params()
    for x in (p, po):
        x.max_iterations, x.max_correspondence_distance, x.ransac_iterations = 10, 25.0, 0
    rc, T, out, st = capi.icp_align_point_to_plane(tree, src, tn, p)
    rco, To, outo, sto = orc.icp_align_point_to_plane(src, tgt, tn, po, nthreads=16)
    assert rc == 0 and rco == 0 and st.iterations == sto.iterations == 10
    assert st.last_correspondences == sto.last_correspondences
    assert np.abs(T - To).max() <= 1e-5, np.abs(T - To).max()
    # how far BOTH end from the motion the source was made with: the linearised point-to-plane estimator on a cloud of
    # 64 planes leaves the in-plane components weakly determined (bench.py reports the same figure for the GPU loop)
    assert abs(np.abs(T - T0).max() - np.abs(To - T0).max()) <= 2e-5
    tree.close()


@pytest.mark.parametrize("n", [30_000, 1_000_000])
def test_fp32_reference_estimator_is_within_its_accumulation_bound(capi, orc, n):
    # The 1e-5 bar is held against the reference's estimator instantiated with Scalar = double
    # (transformation_estimation_svd.h:57); its DEFAULT instantiation accumulates centroids and H in fp32
    # (centroid.hpp:48-75, transformation_estimation_svd.hpp:114-134).  One iteration from the same pairs: the GPU's
    # fp64 result and the oracle's fp32-estimator result differ by no more than the rounding of sequential fp32 sums -
    # a random walk of n additions, each rounding a partial sum of up to n M / 2 to 2^-24 relative:
    #   |error of a mean| <~ c sqrt(n) 2^-24 M     (M = coordinate magnitude; c = 4 leaves room for the tail)
    # and the rotation's entries by the same figure relative to the cloud's extent.
    tgt, src, T0 = c4_clouds()
    sel = np.random.default_rng(11).choice(len(src), n, replace=False)
    src = np.ascontiguousarray(src[sel])
    tree = capi.Tree(tgt)
    p, po32, po64 = capi.icp_default_params(), orc.icp_default_params(), orc.icp_default_params()
    for x in (p, po32, po64):
        x.max_iterations, x.max_correspondence_distance, x.ransac_iterations = 1, 25.0, 0
    po64.svd_scalar_double = 1
    rc, T, _, st = capi.icp_align(tree, src, p)
    _, T32, _, st32 = orc.icp_align(src, tgt, po32, nthreads=16)
    _, T64, _, st64 = orc.icp_align(src, tgt, po64, nthreads=16)
    assert rc == 0 and st.last_correspondences == st32.last_correspondences == st64.last_correspondences
    assert np.abs(T - T64).max() <= 1e-5
    M, extent, u = 1024.0, 1024.0, 2.0 ** -24
    bound_t = 4 * np.sqrt(n) * u * M
    bound_r = bound_t / extent * 4  # H's entries are sums of products of demeaned coordinates: same walk, twice
    assert np.abs(T[:3, 3] - T32[:3, 3]).max() <= bound_t, (np.abs(T[:3, 3] - T32[:3, 3]).max(), bound_t)
    assert np.abs(T[:3, :3] - T32[:3, :3]).max() <= bound_r, (np.abs(T[:3, :3] - T32[:3, :3]).max(), bound_r)
    tree.close()


def test_c4_shards_of_8_and_4_gpus_resident_loop(capi, orc):
    # configs[3] as rank 4 of 8 (125k source points) and rank 1 of 4 (250k) see it: the shard's iterations run inside
    # ONE launch (every workgroup resident); the state after 12 iterations equals 12 launches of one iteration, bit for
    # bit, and the shards' sums are those of the full cloud's loop restricted to the shard (same pairs per point)
    tgt, src, T0 = c4_clouds()
    tree = capi.Tree(tgt)
    for rank, world in ((4, 8), (1, 4)):
        a = capi.IcpShard(tree, src, rank=rank, world=world)
        b = capi.IcpShard(tree, src, rank=rank, world=world)
        assert a.shard_size == len(src) // world
        a.run_dev(25.0, 12)
        for _ in range(12):
            b.run_dev(25.0, 1)
        rca, Ta, sta = a.state()
        rcb, Tb, stb = b.state()
        assert rca == rcb == 0 and sta.iterations == stb.iterations == 12
        assert sta.last_correspondences == stb.last_correspondences == a.shard_size
        assert np.array_equal(Ta, Tb) and sta.fitness_delta == stb.fitness_delta
        assert np.abs(Ta - T0).max() < 5e-2  # one shard alone already finds the motion of the whole cloud
        a.end(want_aligned=False)
        b.end(want_aligned=False)
    tree.close()
