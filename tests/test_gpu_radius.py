"""GPU parity: pclmi_radius (HIP, through the C ABI) vs the CPU oracle, bit-exact CSR output."""
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


def _same(g, o):
    go, gi, gd = g
    oo, oi, od = o
    assert np.array_equal(go, oo), "offsets differ"
    assert np.array_equal(gi, oi), "indices differ"
    assert np.array_equal(gd.view(np.int32), od.view(np.int32)), "distances differ"


@pytest.mark.parametrize("r", [0.01, 0.02, 0.04, 0.08])
def test_uniform_1200_radii(capi, orc, r):
    # recipe and radii of test/test_search.cpp:571-617,624-631
    pts = util.uniform_cloud(1200, 7)
    q = util.uniform_cloud(333, 8)
    t = capi.Tree(pts)
    for max_nn, srt in ((0, True), (0, False), (5, True), (5, False), (1, True), (5000, True)):
        _same(t.radius(q, r, max_nn, srt), orc.radius_brute(pts, q, r, max_nn, srt))


def test_grid_strict_radius_and_ties(capi, orc):
    # 11^3 grid, radius 0.15 around grid points (test/kdtree/test_kdtree.cpp:100-131): strict <,
    # many exactly equal distances -> (d2, index) order
    pts = util.grid_cloud()
    q = np.concatenate([pts[::5], np.array([[0, 0, 0], [0.05, 0.05, 0.05], [10, 10, 10]], np.float32)])
    t = capi.Tree(pts)
    for r in (0.15, 0.1, 0.2, 0.1000001):
        for max_nn, srt in ((0, True), (0, False), (7, True), (7, False)):
            _same(t.radius(q, r, max_nn, srt), orc.radius_brute(pts, q, r, max_nn, srt))
    # the reference test's own check: every result within the radius, count equals brute force
    off, idx, d2 = t.radius(np.zeros((1, 3), np.float32), 0.15)
    brute = np.nonzero((pts * pts).sum(1, dtype=np.float32) < np.float32(0.15 * 0.15))[0]
    assert sorted(idx.tolist()) == brute.tolist()


def test_long_segments_and_everything_in_range(capi, orc):
    # segments beyond the in-LDS sort (1024) and a radius that covers the whole cloud
    pts = util.uniform_cloud(5000, 3)
    q = util.uniform_cloud(40, 4)
    t = capi.Tree(pts)
    for r, max_nn, srt in ((0.5, 0, True), (0.5, 0, False), (3.0, 0, True), (3.0, 2000, False), (0.5, 1500, True),
                           (1e20, 0, True)):
        _same(t.radius(q, r, max_nn, srt), orc.radius_brute(pts, q, r, max_nn, srt))


def test_empty_results_nan_queries_and_errors(capi, orc):
    pts = util.uniform_cloud(3000, 5)
    q = util.uniform_cloud(200, 6)
    q[3] = np.nan
    q[150, 0] = np.inf
    t = capi.Tree(pts)
    off, idx, d2 = t.radius(q, 0.05)
    good = np.isfinite(q).all(axis=1)
    oo, oi, od = orc.radius_brute(pts, q[good], 0.05)
    cnt = np.diff(off)
    assert cnt[3] == 0 and cnt[150] == 0
    assert np.array_equal(cnt[good], np.diff(oo)) and np.array_equal(idx, oi) and np.array_equal(d2, od)
    # zero / tiny radius: nothing is strictly inside
    off, idx, d2 = t.radius(pts[:50], 0.0)
    assert off.tolist() == [0] * 51 and len(idx) == 0
    off, idx, d2 = t.radius(q[:0], 0.1)
    assert off.tolist() == [0] and len(idx) == 0
    e = capi.Tree(np.zeros((0, 3), np.float32))
    off, idx, d2 = e.radius(q[:5], 1.0)
    assert off.tolist() == [0] * 6
    # duplicates of the query point itself are at distance 0 < r2
    off, idx, d2 = t.radius(pts[:10], 1e-6)
    assert np.diff(off).tolist() == [1] * 10 and idx.tolist() == list(range(10)) and not d2.any()


@pytest.mark.parametrize("n,nq,mean_m", [(100_000, 20_000, 32), (250_000, 5_000, 200)])
def test_large_vs_oracle_tree(capi, orc, n, nq, mean_m):
    # BASELINE C3 scaled down: uniform cube [0,1024)^3, r tuned to `mean_m` expected neighbours
    pts = util.uniform_cloud(n, 20260101, 0, 1024)
    q = util.uniform_cloud(nq, 20260102, 0, 1024)
    rho = n / 1024.0 ** 3
    r = (mean_m * 3 / (4 * np.pi * rho)) ** (1 / 3)
    t = capi.Tree(pts)
    ot = orc.Tree(pts)
    for max_nn, srt in ((0, True), (0, False), (10, True)):
        g = t.radius(q, r, max_nn, srt)
        _same(g, ot.radius(q, r, max_nn, srt, nthreads=8))
    assert abs(np.diff(g[0]).mean() - 10) < 0.5


def test_surface_like_cloud(capi, orc):
    rng = np.random.default_rng(9)
    n = 60_000
    uv = rng.random((n, 2), dtype=np.float32) * 100
    pts = np.stack([uv[:, 0], uv[:, 1], rng.integers(0, 4, n) * 7 + rng.normal(0, 0.05, n)], axis=1).astype(np.float32)
    q = np.concatenate([pts[::40] + np.float32(0.2), rng.random((300, 3), dtype=np.float32) * 300 - 100]).astype(np.float32)
    t = capi.Tree(pts)
    _same(t.radius(q, 1.5), orc.Tree(pts).radius(q, 1.5, nthreads=8))


def test_sparse_batches_take_the_per_lane_kernel(capi, orc):
    """Fewer than a quarter as many queries as target points: radius_lane_kernel (count, fill and provisional-row
    modes).  Grid ties at the strict radius, max_nn, unsorted output, more than 64 hits per query (two-traversal
    path), NaN queries - and the same batch embedded in a dense one (union kernel) gives the same rows."""
    pts = util.grid_cloud()  # 1331 points
    q = np.concatenate([pts[::13], pts[:50] + np.float32(0.05)])
    q[3] = np.nan
    assert len(q) * 4 < len(pts)
    t = capi.Tree(pts)
    for r in (0.1, 0.15, 0.31, 2.0):  # 0.31: > 64 hits for inner grid points; 2.0: everything in range
        for max_nn, srt in ((0, True), (0, False), (7, True), (7, False)):
            _same(t.radius(q, r, max_nn, srt), orc.radius_brute(pts, q, r, max_nn, srt))
    big = util.uniform_cloud(150_000, 71, 0, 100)
    tb = capi.Tree(big)
    sparse = util.uniform_cloud(30_001, 72, 0, 100)
    dense = np.concatenate([sparse, util.uniform_cloud(10_000, 73, 0, 100)])  # 40 001 * 4 > 150 000: union kernel
    so, si, sd = tb.radius(sparse, 2.5)
    do, di, dd = tb.radius(dense, 2.5)
    n = int(so[-1])
    assert np.array_equal(so, do[:30_002]) and np.array_equal(si, di[:n]) and np.array_equal(sd.view(np.int32), dd[:n].view(np.int32))
    _same(tb.radius(sparse[:3000], 2.5), orc.Tree(big).radius(sparse[:3000], 2.5, nthreads=8))


def test_a_few_queries_from_the_host(capi, orc):
    """pclmi_radius with up to 1024 queries (pcl::KdTreeFLANN::radiusSearch, one point per call) takes two launches and
    one wait - rows of at most 64 neighbours sorted by one wavefront each, written into pinned host memory.  Same CSR
    output as the oracle in every mode; a query with more than 64 neighbours sends the call to the batch path."""
    pts = util.uniform_cloud(50_000, 91)
    q = util.uniform_cloud(1024, 92)
    q[5] = pts[123]                      # a query on a target point
    q[9] = np.nan                        # and one that is not searchable
    t = capi.Tree(pts)
    r = 0.035                            # ~9 neighbours
    for nq in (1, 2, 17, 64, 65, 300, 1024, 1025):
        for max_nn, srt in ((0, True), (0, False), (4, True), (4, False), (1, True), (100000, True)):
            if nq > 64 and (max_nn, srt) not in ((0, True), (4, False)):
                continue
            _same(t.radius(q[:nq], r, max_nn, srt), orc.radius_brute(pts, q[:nq], r, max_nn, srt))
    # rows of exactly 64 and of 65+ neighbours: the second kind falls through to the batch path
    d = np.sort(((pts - q[0]) ** 2).sum(1, dtype=np.float32))
    for m in (64, 65, 300):
        rr = float(np.sqrt((np.float64(d[m - 1]) + np.float64(d[m])) / 2))
        off, idx, d2 = t.radius(q[:3], rr)
        assert off[1] - off[0] == m
        _same((off, idx, d2), orc.radius_brute(pts, q[:3], rr))
    # nothing in range, strided (16-byte) query rows
    off, idx, d2 = t.radius(q[:4] + np.float32(50), 0.01)
    assert off.tolist() == [0] * 5 and len(idx) == 0
    q4 = np.concatenate([q[:8], np.ones((8, 1), np.float32)], axis=1)
    _same(t.radius(q4, r), orc.radius_brute(pts, q[:8], r))


def test_enqueue_only_radius_search(capi, orc):
    """pclmi_radius_dev_async: caller-owned result arrays of a given capacity, nothing waits for the host.  Same CSR
    output as the synchronous call; the status pair flags a row of more than 64 neighbours or a capacity that is too
    small (the offsets are valid either way); two calls in flight on two streams do not disturb each other."""
    import torch
    pts = util.uniform_cloud(200_000, 93)
    q = util.uniform_cloud(100_000, 94)
    q[7] = np.nan
    t = capi.Tree(pts)
    r = 0.022                                     # ~9 neighbours, at most a few dozen
    dq = torch.from_numpy(q).cuda()
    nq = len(q)

    def bufs(cap):
        return (torch.empty(nq + 1, dtype=torch.int64, device="cuda"), torch.empty(cap, dtype=torch.int32, device="cuda"),
                torch.empty(cap, dtype=torch.float32, device="cuda"), torch.zeros(2, dtype=torch.int64, device="cuda"))
    for srt in (True, False):
        ref = t.radius(q, r, 0, srt)
        total = int(ref[0][-1])
        off, idx, d2, st = bufs(total + 1000)
        t.radius_dev_async(dq.data_ptr(), nq, 12, r, off.data_ptr(), idx.data_ptr(), d2.data_ptr(), total + 1000,
                           st.data_ptr(), sorted_results=srt)
        torch.cuda.synchronize()
        assert st.tolist() == [total, 0]
        _same((off.cpu().numpy(), idx[:total].cpu().numpy(), d2[:total].cpu().numpy()), ref)
    # capacity one short: flagged, offsets still right
    off2, idx2, d22, st2 = bufs(total - 1)
    t.radius_dev_async(dq.data_ptr(), nq, 12, r, off2.data_ptr(), idx2.data_ptr(), d22.data_ptr(), total - 1, st2.data_ptr())
    torch.cuda.synchronize()
    assert st2.tolist() == [total, 1] and np.array_equal(off2.cpu().numpy(), ref[0])
    # a radius with more than 64 neighbours for some query: flagged
    big = t.radius(q[:1000], 0.08)
    assert np.diff(big[0]).max() > 64
    off3, idx3, d23, st3 = bufs(int(big[0][-1]) + 10)
    t.radius_dev_async(dq.data_ptr(), 1000, 12, 0.08, off3.data_ptr(), idx3.data_ptr(), d23.data_ptr(),
                       int(big[0][-1]) + 10, st3.data_ptr())
    torch.cuda.synchronize()
    assert st3.tolist() == [int(big[0][-1]), 1] and np.array_equal(off3[:1001].cpu().numpy(), big[0])
    # two calls in flight on two streams, different radii
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    refs = [t.radius(q, rr) for rr in (0.02, 0.025)]
    outs = [bufs(int(rf[0][-1])) for rf in refs]
    torch.cuda.synchronize()
    for rep in range(3):
        for (rr, o, rf, ss) in zip((0.02, 0.025), outs, refs, (s1, s2)):
            t.radius_dev_async(dq.data_ptr(), nq, 12, rr, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                               int(rf[0][-1]), o[3].data_ptr(), stream=ss.cuda_stream)
    torch.cuda.synchronize()
    for o, rf in zip(outs, refs):
        assert o[3].tolist() == [int(rf[0][-1]), 0]
        _same((o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy()), rf)
    # nothing to find / nothing to search
    off4, idx4, d24, st4 = bufs(16)
    t.radius_dev_async(dq.data_ptr(), 50, 12, 1e-9, off4.data_ptr(), idx4.data_ptr(), d24.data_ptr(), 16, st4.data_ptr())
    t.radius_dev_async(dq.data_ptr(), 0, 12, 0.1, off4.data_ptr(), idx4.data_ptr(), d24.data_ptr(), 16, st4.data_ptr())
    torch.cuda.synchronize()
    assert st4.tolist() == [0, 0] and off4[0].item() == 0
