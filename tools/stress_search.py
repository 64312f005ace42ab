"""Randomised differential stress of the search paths against the brute-force oracle (development helper, run on the
GPU box: STRESS_SECONDS=120 python tools/stress_search.py): k-NN over the whole range of k (1 .. 1024: one-neighbour,
union, per-lane, tiled and heap kernels), batch sizes either side of every dispatch rule, clouds with ties (lattice,
duplicates), NaN rows and NaN queries, index views; radius search with and without max_nn / sorting; the
device-pointer entry points (async, pipelined, allocation-free) against the host call; correspondences."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pcl_amd import capi
from oracle import pcl_oracle as orc
import bench

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
t_end = time.time() + float(os.environ.get("STRESS_SECONDS", "90"))
cases = {"knn": 0, "knn_dev": 0, "radius": 0, "corr": 0}


def cloud(n):
    kind = int(rng.integers(5))
    if kind == 0:
        return rng.uniform(0, 64, (n, 3)).astype(np.float32)
    if kind == 1:
        return bench.surface_cloud(max(n, 50), int(rng.integers(1 << 30)))[:n]
    if kind == 2:  # lattice: distance ties everywhere
        return rng.integers(0, 9, (n, 3)).astype(np.float32)
    if kind == 3:  # duplicates of a few points
        base = rng.uniform(0, 16, (max(1, n // 7), 3)).astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    p = rng.normal(0, 5, (n, 3)).astype(np.float32)  # clustered, one axis flat
    p[:, int(rng.integers(3))] = np.float32(1.25)
    return p


def fail(msg, **arrs):
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/stress_search_fail.npz", **arrs)
    print("MISMATCH", msg)
    sys.exit(1)


while time.time() < t_end:
    n = int(rng.choice([1, 2, 17, 63, 64, 65, 300, 2000, 9000, 40000]))
    n = max(1, n + int(rng.integers(-3, 4)))
    pts = cloud(n)
    n = len(pts)
    if rng.random() < 0.3 and n > 4:
        pts[rng.integers(0, n, max(1, n // 13))] = np.nan
    valid = np.nonzero(np.isfinite(pts).all(axis=1))[0]
    if len(valid) == 0:
        continue
    tree = capi.Tree(pts)
    assert len(tree) == len(valid)
    vp = pts[valid]
    # ---- k-NN from the host
    nq = int(rng.choice([1, 3, 64, 65, 700, 4096, 4097, 16384, 20000]))
    nq = min(nq, max(1, 60_000_000 // (len(valid) * 8)))  # keep the brute-force check to a few seconds
    q = (vp[rng.integers(0, len(vp), nq)] + rng.normal(0, float(rng.choice([0.0, 0.3, 4.0])), (nq, 3))).astype(np.float32)
    if rng.random() < 0.3:
        q[rng.integers(0, nq, max(1, nq // 50))] = np.nan
    good = np.isfinite(q).all(axis=1)
    k = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 32, 33, 64, 65, 100, 128, 129, 300, 512, 1024]))
    if k > 64:
        nq = min(nq, max(1, 4_000_000 // (len(valid) + 1)) + 1)
        q, good = q[:nq], good[:nq]
    gi, gd = tree.knn(q, k)
    oi, od = orc.knn_brute(vp, q[good], k)
    if not (np.array_equal(gi[good], valid[oi]) and np.array_equal(gd[good].view(np.int32), od.view(np.int32))):
        fail("knn n %d nq %d k %d" % (n, nq, k), pts=pts, q=q, k=k)
    if not ((gi[~good] == -1).all() and np.isnan(gd[~good]).all()):
        fail("knn NaN rows n %d nq %d k %d" % (n, nq, k), pts=pts, q=q, k=k)
    cases["knn"] += 1
    # ---- device-pointer entry points against the host call (k <= 64: the allocation-free one's range)
    ke = min(k, len(valid))
    if ke == k and k <= 64:
        stride = int(rng.choice([12, 16, 32]))
        qd = torch.zeros(nq, stride // 4, dtype=torch.float32, device="cuda")
        qd[:, :3] = torch.from_numpy(q).cuda()
        for name in ("knn_dev", "knn_dev_async", "knn_dev_pipelined", "knn_dev_noalloc"):
            oi_d = torch.full((nq, ke), -7, dtype=torch.int32, device="cuda")
            od_d = torch.full((nq, ke), -7.0, dtype=torch.float32, device="cuda")
            if name == "knn_dev_noalloc" and k > 32:
                continue
            torch.cuda.synchronize()  # the library's streams do not wait for torch's: the fills above first
            getattr(tree, name)(qd.data_ptr(), nq, stride, k, oi_d.data_ptr(), od_d.data_ptr())
            torch.cuda.synchronize()
            if not (np.array_equal(oi_d.cpu().numpy(), gi) and np.array_equal(od_d.cpu().numpy(), gd, equal_nan=True)):
                fail("%s n %d nq %d k %d stride %d" % (name, n, nq, k, stride), pts=pts, q=q, k=k)
        cases["knn_dev"] += 1
    # ---- radius
    nq = min(nq, 3000)
    q = q[:nq][np.isfinite(q[:nq]).all(axis=1)]
    if len(q):
        ext = float(np.ptp(vp, axis=0).max()) + 1e-3
        r = float(rng.choice([0.0, 1e-3, 0.02, 0.1, 0.5, 3.0])) * ext
        if r * r * 1.0 == 0.0:
            r = 1e-6
        mx, srt = int(rng.choice([0, 0, 1, 3, 40, 100])), bool(rng.integers(2))
        g = tree.radius(q, r, mx, srt)
        o = orc.radius_brute(vp, q, r, mx, True)
        same = np.array_equal(g[0], o[0])
        if same and (srt or mx > 0):
            # (max_nn > 0 keeps the NEAREST max_nn: a set; unsorted output is compared as sets per query below)
            pass
        if not same:
            fail("radius offsets n %d nq %d r %g max_nn %d sorted %d" % (n, len(q), r, mx, srt), pts=pts, q=q, r=r, mx=mx)
        if srt:
            if not (np.array_equal(g[1], valid[o[1]]) and np.array_equal(g[2].view(np.int32), o[2].view(np.int32))):
                fail("radius rows n %d nq %d r %g max_nn %d" % (n, len(q), r, mx), pts=pts, q=q, r=r, mx=mx)
        else:
            for i in rng.integers(0, len(q), 50):
                a, b = int(g[0][i]), int(g[0][i + 1])
                if mx == 0 or b - a < mx:  # with ties at the cut the unsorted set is not unique
                    if sorted(g[1][a:b].tolist()) != sorted(valid[o[1][a:b]].tolist()):
                        fail("radius set n %d r %g max_nn %d" % (n, r, mx), pts=pts, q=q, r=r, mx=mx)
        cases["radius"] += 1
    # ---- correspondences (1-NN + distance cut) on the finite cloud
    if 3 <= len(vp) <= 20000:
        src = (vp[rng.integers(0, len(vp), min(len(vp), 5000))] + rng.normal(0, 0.2, (min(len(vp), 5000), 3))).astype(np.float32)
        t2 = capi.Tree(vp)
        md = float(rng.choice([0.05, 0.5, 1e9]))
        gc = capi.correspondences(t2, src, md)
        oc = orc.correspondences(src, vp, md)
        if not (len(gc) == len(oc) and np.array_equal(gc["index_query"], oc["index_query"]) and
                np.array_equal(gc["distance"].view(np.int32), oc["distance"].view(np.int32)) and
                np.array_equal(gc["index_match"], oc["index_match"])):
            fail("correspondences n %d md %g" % (len(vp), md), pts=vp, q=src, md=md)
        t2.close()
        cases["corr"] += 1
    tree.close()
print("stress_search OK:", cases)
