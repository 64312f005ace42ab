"""GPU parity of the registration path (through the C ABI) vs the golden vectors and the CPU oracle.

Bars: indices and fp32 distances / transformed points bit-exact; 4x4 transforms within 1e-5 of the
oracle (north_star) and within the reference tests' own tolerances of the golden matrices.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

T_TOL = 1e-5  # north_star: final ICP transforms within 1e-5


def f32_sum_tol(n, scale):
    """Error the reference's fp32 instantiation itself carries: its centroids are running fp32 sums
    of n coordinates of magnitude `scale` (centroid.hpp:48-75), a random walk of ~sqrt(n) roundings of
    half an ulp of the running sum (n * scale), divided by n."""
    return 4 * np.sqrt(n) * np.finfo(np.float32).eps * scale


@pytest.fixture(scope="module")
def capi():
    from pcl_amd import capi as c
    assert c.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return c


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


def rigid(angle_deg, axis, trans):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    th = np.deg2rad(angle_deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = trans
    return T.astype(np.float32)


def surface_cloud(n, seed):
    """Points on a wavy sheet + a few planes: a 2-manifold-like cloud (SURVEY 8d surface variant)."""
    rng = np.random.default_rng(seed)
    uv = rng.random((n, 2), dtype=np.float32) * 100
    z = 5 * np.sin(uv[:, 0] * 0.1) * np.cos(uv[:, 1] * 0.07) + rng.integers(0, 3, n) * 20
    return np.stack([uv[:, 0], uv[:, 1], z.astype(np.float32)], axis=1).astype(np.float32)


def test_transform_bit_exact(capi, orc):
    pts = util.uniform_cloud(5000, 3, -50, 50)
    T = rigid(33, (1, 2, 3), (4, -5, 6))
    g = capi.transform(pts, T)
    o = orc.transform(pts, T)
    assert np.array_equal(g.view(np.int32), o.view(np.int32))
    # non-finite rows are left untouched (transforms.hpp:196-204)
    pts[7] = np.nan
    pts[11, 2] = np.inf
    g = capi.transform(pts, T)
    assert np.isnan(g[7]).all() and np.array_equal(g[11], pts[11])
    # 16-byte PointXYZ rows keep their 4th float
    p4 = np.concatenate([pts[:100], np.full((100, 1), 1, np.float32)], axis=1)
    g4 = capi.transform(p4, T)
    assert np.array_equal(g4[:, 3], p4[:, 3]) and np.array_equal(g4[:20, :3], g[:20], equal_nan=True)


def test_bun_correspondences_golden(capi, orc):
    api = util.golden("registration_api.json")
    src, tgt = util.bun0(), util.bun4()
    tt, ts = capi.Tree(tgt), capi.Tree(src)
    corr = capi.correspondences(tt, src)
    exp = np.array(api["correspondences_original"])
    assert corr["index_query"].tolist() == exp[:, 0].tolist()
    assert corr["index_match"].tolist() == exp[:, 1].tolist()
    oc = orc.correspondences(src, tgt)
    assert np.array_equal(corr["distance"].view(np.int32), oc["distance"].view(np.int32))
    rec = capi.correspondences_reciprocal(tt, ts, src, tgt)
    exp = np.array(api["correspondences_reciprocal"])
    assert rec["index_query"].tolist() == exp[:, 0].tolist()
    assert rec["index_match"].tolist() == exp[:, 1].tolist()
    # max distance is a non-strict keep (correspondence_estimation.hpp:108)
    md = 0.01
    c2 = capi.correspondences(tt, src, md)
    o2 = orc.correspondences(src, tgt, md)
    assert np.array_equal(c2, o2)
    # an index view of the source (PCLBase indices_)
    view = np.arange(0, 397, 3, dtype=np.int32)
    c3 = capi.correspondences(tt, src, src_indices=view)
    assert c3["index_query"].tolist() == view.tolist()
    assert c3["index_match"].tolist() == exp_match(api, view)


def exp_match(api, view):
    exp = np.array(api["correspondences_original"])
    return exp[view, 1].tolist()


def test_bun_transformation_estimation_svd(capi, orc):
    api = util.golden("registration_api.json")
    src, tgt = util.bun0(), util.bun4()
    rec = orc.correspondences(src, tgt, reciprocal=True)
    T = capi.estimate_rigid_svd(src, tgt, rec["index_query"], rec["index_match"])
    assert np.allclose(T, np.array(api["transform_from_SVD"]), atol=1e-4)  # the reference test's bar
    To = orc.estimate_rigid_svd(src, tgt, rec["index_query"], rec["index_match"])
    assert np.abs(T - To).max() <= T_TOL
    assert T[3].tolist() == [0, 0, 0, 1]
    # recover a known rigid motion from exact pairs
    pts = util.uniform_cloud(20000, 5, -10, 10)
    Tk = rigid(17, (0.3, -1, 2), (1, 2, -3))
    moved = orc.transform(pts, Tk)
    Te = capi.estimate_rigid_svd(pts, moved)
    assert np.abs(Te - Tk).max() < 2e-5


@pytest.mark.parametrize("ransac", [0, 1000])
def test_bun_full_icp_golden(capi, orc, ransac):
    # test/registration/test_registration.cpp:154-188
    g = util.golden("icp_bun0_bun4.json")
    src, tgt = util.bun0(), util.bun4()
    tt = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.max_iterations = g["max_iterations"]
    p.transformation_epsilon = g["transformation_epsilon"]
    p.max_correspondence_distance = g["max_correspondence_distance"]
    p.ransac_iterations = ransac
    rc, T, out, st = capi.icp_align(tt, src, p)
    assert rc == 0 and st.converged
    gold, tol = np.array(g["matrix"]), np.array(g["tol"])
    assert np.all(np.abs(T[:3] - gold[:3]) <= tol[:3]), (T, gold)
    assert T[3].tolist() == [0, 0, 0, 1]
    po = orc.icp_default_params()
    po.max_iterations, po.transformation_epsilon = p.max_iterations, p.transformation_epsilon
    po.max_correspondence_distance, po.ransac_iterations = p.max_correspondence_distance, ransac
    rco, To, outo, sto = orc.icp_align(src, tgt, po)
    # transformation_epsilon = 1e-8: the fp64 solve reaches the fixed point exactly (T == T_prev) and
    # stops, the reference's fp32 solve keeps jittering by ~1e-7 until max_iterations - same fixed point
    assert st.iterations <= sto.iterations
    assert st.last_correspondences == sto.last_correspondences
    assert np.abs(T - To).max() <= T_TOL, np.abs(T - To).max()
    assert np.allclose(out[:, :3], outo[:, :3], atol=2e-5)
    assert np.array_equal(out[:, 3], np.ones(397, np.float32))
    # getFitnessScore (registration.hpp:71-109)
    fg = capi.fitness_score(tt, src, T)
    fo = orc.fitness_score(src, tgt, T)
    assert abs(fg - fo) <= 1e-12 + 1e-9 * abs(fo)


@pytest.mark.parametrize("n,ransac,maxd", [(30000, 0, 3.0), (30000, 1000, 3.0), (200_000, 0, 3.0),
                                          (30000, 1000, 1e3)])  # 1e3: every source point keeps its pair
def test_synthetic_icp_vs_oracle(capi, orc, n, ransac, maxd):
    tgt = surface_cloud(n, 41)
    rng = np.random.default_rng(42)
    sel = rng.permutation(n)[: (n * 9) // 10]
    T0 = rigid(2.0, (1, 1, 1), (0.4, -0.2, 0.1))
    src = orc.transform(tgt[sel] + rng.normal(0, 0.01, (len(sel), 3)).astype(np.float32), np.linalg.inv(T0))
    tt = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.max_iterations = 12
    p.max_correspondence_distance = maxd
    p.ransac_iterations = ransac
    p.ransac_outlier_threshold = 0.5
    rc, T, out, st = capi.icp_align(tt, src, p)
    po = orc.icp_default_params()
    po.max_iterations, po.max_correspondence_distance = 12, maxd
    po.ransac_iterations, po.ransac_outlier_threshold = ransac, 0.5
    # the 1e-5 bar is against the reference estimator instantiated with Scalar = double
    # (transformation_estimation_svd.h:57); its fp32 instantiation is itself only accurate to
    # f32_sum_tol at this size (on the reference's own bun fixture both agree to 1e-5, see above)
    po.svd_scalar_double = 1
    rco, To, outo, sto = orc.icp_align(src, tgt, po, nthreads=8)
    assert rc == 0 and rco == 0
    assert st.iterations == sto.iterations == 12
    assert st.last_correspondences == sto.last_correspondences
    assert maxd < 100 or st.last_correspondences == len(src)
    assert np.abs(T - To).max() <= T_TOL, np.abs(T - To).max()
    po.svd_scalar_double = 0
    rco, To32, outo, sto = orc.icp_align(src, tgt, po, nthreads=8)
    assert np.abs(T - To32).max() <= f32_sum_tol(len(src), 100.0), np.abs(T - To32).max()
    if n <= 30000:
        assert np.abs(T - T0).max() < 5e-3  # and it actually registers the clouds


@pytest.mark.parametrize("scale,ransac", [((1.0, 1.0, 1.0), 0), ((0.5, 2.0, 0.25), 0), ((0.5, 2.0, 0.25), 1000),
                                          ((1.0, 1.0, 0.0), 0)])
def test_icp_with_a_rescaling_point_representation(capi, orc, scale, ransac):
    """Registration::setPointRepresentation (registration.h:362-370): searches in scale * (x, y, z) of the MOVING
    source point, estimator / RANSAC / transform on the points themselves.  (1, 1, 0) is CustomPointRepresentation(2, 0):
    the search ignores z."""
    n = 30000
    tgt = surface_cloud(n, 141)
    rng = np.random.default_rng(142)
    sel = rng.permutation(n)[: (n * 9) // 10]
    T0 = rigid(1.5, (1, 0.5, 1), (0.3, -0.2, 0.1))
    src = orc.transform(tgt[sel] + rng.normal(0, 0.01, (len(sel), 3)).astype(np.float32), np.linalg.inv(T0))
    sc = np.asarray(scale, np.float32)
    tt = capi.Tree(tgt * sc)  # what KdTreeFLANN::setInputCloud builds under that representation
    p = capi.icp_default_params()
    p.max_iterations, p.max_correspondence_distance = 10, 2.0
    p.ransac_iterations, p.ransac_outlier_threshold = ransac, 0.5
    rc, T, out, st = capi.icp_align_scaled(tt, sc, tgt, src, p)
    po = orc.icp_default_params()
    po.max_iterations, po.max_correspondence_distance = 10, 2.0
    po.ransac_iterations, po.ransac_outlier_threshold = ransac, 0.5
    po.svd_scalar_double = 1
    rco, To, outo, sto = orc.icp_align_scaled(sc, src, tgt, po, nthreads=8)
    assert rc == 0 and rco == 0
    assert st.iterations == sto.iterations == 10
    assert st.last_correspondences == sto.last_correspondences
    assert np.abs(T - To).max() <= T_TOL, np.abs(T - To).max()
    assert np.allclose(out[:, :3], outo[:, :3], atol=1e-4)
    if scale == (1.0, 1.0, 1.0):  # the trivial representation: the plain entry point, bit for bit
        rc1, T1, out1, st1 = capi.icp_align(capi.Tree(tgt), src, p)
        assert np.array_equal(T1, T) and np.array_equal(out1, out, equal_nan=True)
    else:  # and the rescale changes the pairing: not the unscaled result
        rc1, T1, out1, st1 = capi.icp_align(capi.Tree(tgt), src, p)
        assert st1.last_correspondences != st.last_correspondences or not np.array_equal(T1, T)
    # a guess, and a context that already stepped refuses the switch
    rc, Tg, _, _ = capi.icp_align_scaled(tt, sc, tgt, src, p, guess=T0)
    rco, Tgo, _, _ = orc.icp_align_scaled(sc, src, tgt, po, guess=T0, nthreads=8)
    assert rc == 0 and np.abs(Tg - Tgo).max() <= T_TOL


def test_icp_guess_threshold_and_not_enough(capi, orc):
    tgt = surface_cloud(20000, 51)
    T0 = rigid(3.0, (0, 0, 1), (0.5, 0.5, 0))
    src = orc.transform(tgt[::2], np.linalg.inv(T0))
    tt = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.ransac_iterations = 0
    p.max_iterations = 5
    p.max_correspondence_distance = 2.0
    guess = rigid(2.5, (0, 0, 1), (0.4, 0.4, 0))
    rc, T, out, st = capi.icp_align(tt, src, p, guess=guess)
    po = orc.icp_default_params()
    po.ransac_iterations, po.max_iterations, po.max_correspondence_distance = 0, 5, 2.0
    po.svd_scalar_double = 1
    rco, To, outo, sto = orc.icp_align(src, tgt, po, guess=guess, nthreads=8)
    assert rc == 0 and np.abs(T - To).max() <= T_TOL
    # transformation_epsilon stops early (icp.hpp:194)
    p.max_iterations, po.max_iterations = 100, 100
    p.transformation_epsilon = po.transformation_epsilon = 1e-4
    rc, T, out, st = capi.icp_align(tt, src, p, guess=guess)
    rco, To, outo, sto = orc.icp_align(src, tgt, po, guess=guess, nthreads=8)
    assert st.iterations == sto.iterations < 100 and st.converged
    # far-away source: fewer than 3 pairs under the threshold (icp.hpp:103-108)
    far = src + np.float32(1e4)
    p.max_correspondence_distance = 1.0
    rc, T, out, st = capi.icp_align(tt, far, p)
    assert rc == 6 and not st.converged


def test_sharded_steps_equal_single_shard(capi, orc):
    """Two shards + summed fp64 moments + the shared host solve == one shard."""
    tgt = surface_cloud(50000, 61)
    T0 = rigid(1.5, (1, 0, 1), (0.3, 0.1, -0.2))
    src = orc.transform(tgt[::3], np.linalg.inv(T0))
    tt = capi.Tree(tgt)
    one = capi.IcpShard(tt, src)
    half = len(src) // 2
    a, b = capi.IcpShard(tt, src[:half]), capi.IcpShard(tt, src[half:])
    for it in range(4):
        s1 = one.step(2.0)
        s2 = a.step(2.0) + b.step(2.0)
        assert s1[0] == s2[0] and s1[18] == s2[18]  # pair and point counts are exact
        assert np.allclose(s1[:19], s2[:19], rtol=1e-12, atol=1e-9)
        T1, T2 = capi.icp_solve(s1), capi.icp_solve(s2)
        assert np.abs(T1 - T2).max() <= 1e-6
        one.set_transform(T1)
        a.set_transform(T1)
        b.set_transform(T1)
    o1 = one.end()
    oa, ob = a.end(), b.end()
    assert np.array_equal(o1, np.concatenate([oa, ob]))


def test_sharded_driver_single_rank_equals_icp_align(capi, orc):
    """pcl_amd/sharded.py on one rank (no process group) == pclmi_icp_align with RANSAC off."""
    from pcl_amd import sharded
    src, tgt = util.bun0(), util.bun4()
    tt = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 20, 0.05, 0
    rc, T, out, st = capi.icp_align(tt, src, p)
    rc2, T2, out2, it, ok, pairs = sharded.icp_align_gpu_sharded(tt, src, p)
    assert rc == rc2 == 0 and ok and it == st.iterations and pairs == st.last_correspondences
    assert np.array_equal(T, T2) and np.array_equal(out, out2)


def test_device_resident_loop_equals_host_driven_loop(capi, orc):
    """pclmi_icp_solve_dev / run_dev (solve as a kernel, no host round trip) == the host-driven loop
    (pclmi_icp_step + pclmi_icp_solve + pclmi_icp_set_transform), bit for bit."""
    from pcl_amd import sharded
    tgt = surface_cloud(60000, 71)
    T0 = rigid(1.5, (1, 0, 1), (0.3, 0.1, -0.2))
    src = orc.transform(tgt[::2], np.linalg.inv(T0))
    tt = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.ransac_iterations = 0
    for max_it, teps, guess in [(7, 0.0, None), (60, 1e-5, rigid(1.0, (1, 0, 1), (0.2, 0.1, -0.1)))]:
        p.max_iterations, p.transformation_epsilon, p.max_correspondence_distance = max_it, teps, 2.0
        rc1, T1, out1, it1, ok1, pairs1 = sharded.icp_align_gpu_sharded(tt, src, p, guess=guess)
        rc2, T2, out2, it2, ok2, pairs2 = sharded.icp_align_gpu_sharded_dev(tt, src, p, guess=guess, check_every=3)
        rc3, T3, out3, st3 = capi.icp_align(tt, src, p, guess=guess)
        assert rc1 == rc2 == rc3 == 0 and ok1 and ok2 and st3.converged
        assert it1 == it2 == st3.iterations and pairs1 == pairs2 == st3.last_correspondences
        if teps:
            assert it1 < max_it
        assert np.array_equal(T1, T2) and np.array_equal(T1, T3)
        assert np.array_equal(out1, out2) and np.array_equal(out1, out3)
    # raw stepping without criteria: run_dev never stops by itself; state can be read any time
    ctx = capi.IcpShard(tt, src)
    ctx.run_dev(2.0, 5)
    rc, Ta, st = ctx.state()
    assert rc == 0 and st.iterations == 5 and not st.converged
    ctx.run_dev(2.0, 2)
    rc, Tb, st = ctx.state()
    assert st.iterations == 7
    p.max_iterations, p.transformation_epsilon = 7, 0.0
    rc1, T1, out1, it1, ok1, pairs1 = sharded.icp_align_gpu_sharded(tt, src, p)
    assert np.array_equal(Tb, T1) and np.array_equal(ctx.end(), out1)
    # not enough correspondences is latched on the device and reported by the state read
    ctx = capi.IcpShard(tt, src + np.float32(1e4))
    p.max_correspondence_distance = 1.0
    ctx.set_criteria(p)
    ctx.run_dev(1.0, 4)
    rc, T, st = ctx.state()
    assert rc == 6 and not st.converged and st.iterations == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))
    ctx.end(want_aligned=False)
    # a context is driven from the host or from the device, not both
    ctx = capi.IcpShard(tt, src)
    ctx.run_dev(2.0, 1)
    with pytest.raises(capi.PclmiError):
        ctx.set_transform(np.eye(4, dtype=np.float32))
    ctx.end(want_aligned=False)


@pytest.mark.parametrize("plane", [False, True])
def test_resident_loop_equals_one_launch_per_iteration(capi, orc, plane):
    """pclmi_icp_run_dev with several iterations on a shard whose workgroups are all resident at once runs them inside
    ONE launch (the workgroups wait for the solve between iterations); one iteration per call is one launch each.
    Same arithmetic in the same order: state, pairs and the moved cloud are equal bit for bit; the criteria stop the
    launch from inside."""
    tgt = surface_cloud(60000, 171)
    T0 = rigid(1.5, (1, 0, 1), (0.3, 0.1, -0.2))
    src = orc.transform(tgt[::2] + np.float32(0.01), np.linalg.inv(T0))
    src[5] = np.nan          # rows that are not searchable stay as they are and never pair (transforms.hpp:196-204)
    src[777, 1] = np.inf
    tt = capi.Tree(tgt)
    tn = capi.normals(tt, tgt, k=12)[:, :3].copy() if plane else None

    def begin():
        c = capi.IcpShard(tt, src)
        if plane:
            c.set_point_to_plane(tn)
        return c
    a, b = begin(), begin()
    a.run_dev(2.0, 9)
    for _ in range(9):
        b.run_dev(2.0, 1)
    rca, Ta, sta = a.state()
    rcb, Tb, stb = b.state()
    assert rca == rcb == 0 and sta.iterations == stb.iterations == 9
    assert sta.last_correspondences == stb.last_correspondences and sta.fitness_delta == stb.fitness_delta
    assert np.array_equal(Ta, Tb) and np.array_equal(np.asarray(sta.last_transform), np.asarray(stb.last_transform))
    a.run_dev(2.0, 3)   # a second resident launch carries on from the first
    b.run_dev(2.0, 3)   # (3 per-iteration launches, or one resident launch - either way the same state)
    rca, Ta, sta = a.state()
    rcb, Tb, stb = b.state()
    assert sta.iterations == stb.iterations == 12 and np.array_equal(Ta, Tb)
    assert np.array_equal(a.end(), b.end(), equal_nan=True)
    # criteria: the launch ends itself at max_iterations / the epsilon, later launches are no-ops
    p = capi.icp_default_params()
    p.ransac_iterations, p.max_iterations, p.transformation_epsilon = 0, 40, 1e-6
    a, b = begin(), begin()
    a.set_criteria(p)
    b.set_criteria(p)
    a.run_dev(2.0, 64)
    for _ in range(45):
        b.run_dev(2.0, 1)
    a.run_dev(2.0, 5)
    rca, Ta, sta = a.state()
    rcb, Tb, stb = b.state()
    assert sta.converged and stb.converged and sta.iterations == stb.iterations < 40
    assert np.array_equal(Ta, Tb) and np.array_equal(a.end(), b.end(), equal_nan=True)
    # not enough correspondences inside a resident launch
    c = capi.IcpShard(tt, src + np.float32(1e4))
    p.max_correspondence_distance = 1.0
    c.set_criteria(p)
    c.run_dev(1.0, 6)
    rc, T, st = c.state()
    assert rc == 6 and st.iterations == 0
    c.end(want_aligned=False)


def test_resident_loops_of_one_process_share_the_chip(capi, orc):
    """Resident launches wait for their own workgroups, so those of ONE process must fit on the chip together: a
    context reserves its workgroups (430 each here, the chip holds 1024), the third context in flight does not get them
    and runs one launch per iteration.  All finish, all give the solo result."""
    tgt = surface_cloud(220_000, 181)
    T0 = rigid(1.0, (1, 0, 1), (0.2, 0.1, -0.1))
    src = orc.transform(tgt[::2] + np.float32(0.01), np.linalg.inv(T0))
    tt = capi.Tree(tgt)
    solo = capi.IcpShard(tt, src)
    solo.run_dev(2.0, 8)
    rc0, T_solo, st0 = solo.state()
    out_solo = solo.end()
    ctxs = [capi.IcpShard(tt, src) for _ in range(3)]
    for c in ctxs:       # three launches in flight on three streams
        c.run_dev(2.0, 8)
    for c in ctxs:
        rc, T, st = c.state()
        assert rc == 0 and st.iterations == 8 and np.array_equal(T, T_solo)
    for c in ctxs:       # and again: the reservations were given back by the state reads
        c.run_dev(2.0, 2)
    outs = []
    for c in ctxs:
        rc, T, st = c.state()
        assert st.iterations == 10
        outs.append(c.end())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert st0.iterations == 8 and out_solo.shape == outs[0].shape


def test_rccl_all_reduce_between_the_library_kernels():
    """tests/rccl_single_rank.py in a process of its own: a one-rank nccl (= RCCL) process group, the sharded
    device-resident driver and bench.py's N > 1 loop shape with the all-reduce issued between the library's step and
    solve kernels, equal to pclmi_icp_align bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_single_rank.py"), "29655"],
                       capture_output=True, text=True, timeout=110)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "RCCL single-rank loop OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("ransac,plane", [(0, False), (1000, False), (0, True), (1000, True)])
def test_observed_align_reports_every_iteration(capi, orc, ransac, plane):
    """pclmi_icp_align_observed (Registration::registerVisualizationCallback, icp.hpp:182-183): one call per iteration
    with the moved source and the pairs given to the estimator; the align itself is the unobserved one, bit for bit."""
    tgt = surface_cloud(20000, 77)
    T0 = rigid(1.5, (1, 0, 1), (0.3, 0.1, -0.2))
    src = orc.transform(tgt[::3], np.linalg.inv(T0))
    src[5] = np.nan  # never paired
    tt = capi.Tree(tgt)
    tn = capi.normals(tt, tgt, k=12, viewpoint=(0, 0, 1e4))[:, :3].copy() if plane else None
    p = capi.icp_default_params()
    p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 6, 2.0, ransac
    p.ransac_outlier_threshold = 0.5
    seen = []
    rc, T, out, st = capi.icp_align_observed(tt, src, lambda it, cloud, si, ti: seen.append((it, cloud, si, ti)), p,
                                             tgt_normals=tn)
    if plane:
        rc2, T2, out2, st2 = capi.icp_align_point_to_plane(tt, src, tn, p)
    else:
        rc2, T2, out2, st2 = capi.icp_align(tt, src, p)
    assert rc == rc2 == 0 and st.iterations == st2.iterations == 6
    assert np.array_equal(T, T2) and np.array_equal(out.view(np.int32), out2.view(np.int32))
    assert [s[0] for s in seen] == [1, 2, 3, 4, 5, 6]
    assert np.array_equal(seen[-1][1].view(np.int32), out.view(np.int32))  # the last report is the aligned cloud
    for it, cloud, si, ti in seen:
        assert len(si) == len(ti) > 3 and (np.diff(si) > 0).all() and 5 not in si
        assert si.min() >= 0 and si.max() < len(src) and ti.min() >= 0 and ti.max() < len(tgt)
    assert len(seen[-1][2]) == st.last_inliers
    if ransac == 0:
        # the reported pairs are the nearest neighbours of the cloud as it stood BEFORE this iteration's transform
        prev = seen[-2][1][:, :3]
        idx, d2 = tt.knn(np.nan_to_num(prev), 1)
        keep = (d2[:, 0] < 4.0) & np.isfinite(prev).all(axis=1)
        assert np.array_equal(np.flatnonzero(keep), seen[-1][2]) and np.array_equal(idx[keep, 0], seen[-1][3])


def test_certificates_change_nothing(capi, tmp_path):
    """The step's "same neighbour as last iteration" certificates (TreeView::cert: a point within 0.49 x the distance
    from its previous neighbour to that neighbour's nearest / second nearest other point) replace the walk, never the
    answer: every align of tests/icp_cert_worker.py - clean and noisy surfaces, a target with twins and triplets, a
    lattice of equal distances; RANSAC off and on, the sharded context per launch and as one resident launch,
    point-to-plane - is bit-identical to a process that runs without them (PCLMI_ICP_CERT=0)."""
    import os
    import subprocess
    import sys
    from tests import icp_cert_worker
    here = icp_cert_worker.run(capi)
    path = str(tmp_path / "nocert.npz")
    env = dict(os.environ, PCLMI_ICP_CERT="0")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "icp_cert_worker.py"), path], env=env,
                   check=True, timeout=300)
    there = np.load(path)
    assert set(there.files) == set(here)
    for k in sorted(here):
        assert np.array_equal(np.asarray(here[k]), there[k], equal_nan=True), k
    # the certificates did fire: the clean surface's converged iterations pair every point
    assert here["surface_0_stats"][3] == here["surface_1_stats"][3] > 0
