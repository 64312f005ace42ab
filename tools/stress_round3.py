"""Randomised parity stress of round 3's paths (development helper, ~1-2 minutes on the GPU box): the resident ICP loop
against one launch per iteration, default-parameter ICP (fused mask words + draws, speculative batches) against the
oracle on clouds where not every point finds a pair, the two-launch host radius path and the enqueue-only radius search
against the oracle / the synchronous call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pcl_amd import capi
from oracle import pcl_oracle as orc
import bench
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
t_end = time.time() + float(os.environ.get("STRESS_SECONDS", "90"))
n_cases = [0, 0, 0, 0]
while time.time() < t_end:
    # ---- resident loop
    n = int(rng.integers(1, 260_000))
    tgt = bench.surface_cloud(max(n, 50), int(rng.integers(1 << 30)))
    T0 = bench.rigid(float(rng.uniform(0.2, 2.0)), rng.normal(size=3), rng.normal(size=3) * 0.5)
    sel = rng.permutation(len(tgt))[: max(1, int(len(tgt) * rng.uniform(0.3, 1.0)))]
    src = orc.transform(tgt[sel] + rng.normal(0, 0.05, (len(sel), 3)).astype(np.float32), np.linalg.inv(T0))
    tree = capi.Tree(tgt)
    thr = float(rng.choice([0.5, 2.0, 25.0]))
    a, b = capi.IcpShard(tree, src), capi.IcpShard(tree, src)
    k = int(rng.integers(2, 9))
    a.run_dev(thr, k)
    for _ in range(k):
        b.run_dev(thr, 1)
    ra, Ta, sa = a.state()
    rb, Tb, sb = b.state()
    assert ra == rb and sa.iterations == sb.iterations and sa.last_correspondences == sb.last_correspondences, (n, thr)
    assert np.array_equal(Ta, Tb), (n, thr)
    assert np.array_equal(a.end(), b.end(), equal_nan=True)
    n_cases[0] += 1
    # ---- default-parameter ICP vs the oracle, pairs missing
    if len(src) <= 40_000 and len(src) >= 20:
        p, po = capi.icp_default_params(), orc.icp_default_params()
        # the rejection threshold stays >= 10 sigma of the noise: the reference fits the 3-point model in fp32, the
        # library in fp64, the models differ by ~1e-4 on far-from-origin samples, and at a threshold inside the noise
        # (0.1 at sigma 0.05) a dozen borderline pairs change sides, which moves the refit by ~1e-4
        mcd, rot = float(rng.choice([0.3, 1.0, 5.0])), float(rng.choice([0.5, 2.0]))
        for x in (p, po):
            x.max_iterations = 6
            x.max_correspondence_distance = mcd
            x.ransac_outlier_threshold = rot
        po.svd_scalar_double = 1
        rc, T, out, st = capi.icp_align(tree, src, p)
        rco, To, outo, sto = orc.icp_align(src, tgt, po, nthreads=8)
        assert (rc == 0) == (rco == 0), (rc, rco, len(src))
        # (a few dozen pairs with a third of them inliers: the 3-point models' fp32 / fp64 difference decides single
        #  inliers and the refits part ways - DESIGN 10.9 - although the first iterations agree to the bit)
        if rc == 0 and min(st.last_correspondences, sto.last_correspondences) >= 200 and st.last_inliers >= 100:
            bad = (st.iterations != sto.iterations or st.last_correspondences != sto.last_correspondences or
                   np.abs(T - To).max() > 2e-5 + 1e-6 * float(np.abs(tgt).max()))  # ~8 ulp of the largest coordinate
            if bad:
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez("gpurun_out/stress_fail.npz", tgt=tgt, src=src, mcd=p.max_correspondence_distance,
                         rot=p.ransac_outlier_threshold)
                print("MISMATCH n_src %d mcd %g rot %g: iterations %d / %d, pairs %d / %d, inliers %d, dT %g" % (
                    len(src), p.max_correspondence_distance, p.ransac_outlier_threshold, st.iterations, sto.iterations,
                    st.last_correspondences, sto.last_correspondences, st.last_inliers, np.abs(T - To).max()))
                sys.exit(1)
        n_cases[1] += 1
    # ---- radius: the few-queries host path
    nq = int(rng.integers(1, 1100))
    q = (tgt[rng.integers(0, len(tgt), nq)] + rng.normal(0, 1.0, (nq, 3))).astype(np.float32)
    r = float(rng.uniform(0.5, 6.0))
    mx, srt = int(rng.choice([0, 0, 3, 40])), bool(rng.integers(2))
    g = tree.radius(q, r, mx, srt)
    o = orc.radius_brute(tgt, q, r, mx, srt)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]) and np.array_equal(g[2].view(np.int32), o[2].view(np.int32)), (nq, r, mx, srt)
    n_cases[2] += 1
    # ---- enqueue-only radius vs the synchronous call
    nq = int(rng.integers(1, 60_000))
    q = (tgt[rng.integers(0, len(tgt), nq)] + rng.normal(0, 1.0, (nq, 3))).astype(np.float32)
    ref = tree.radius(q, r, 0, srt)
    total = int(ref[0][-1])
    dq = torch.from_numpy(q).cuda()
    off = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
    oi = torch.empty(total + 8, dtype=torch.int32, device="cuda")
    od = torch.empty(total + 8, dtype=torch.float32, device="cuda")
    stt = torch.zeros(2, dtype=torch.int64, device="cuda")
    tree.radius_dev_async(dq.data_ptr(), nq, 12, r, off.data_ptr(), oi.data_ptr(), od.data_ptr(), total + 8, stt.data_ptr(), sorted_results=srt)
    torch.cuda.synchronize()
    over = bool(np.diff(ref[0]).max() > 64)
    assert stt.tolist() == [total, 1 if over else 0], (stt.tolist(), total, over)
    assert np.array_equal(off.cpu().numpy(), ref[0])
    if not over:
        assert np.array_equal(oi[:total].cpu().numpy(), ref[1]) and np.array_equal(od[:total].cpu().numpy().view(np.int32), ref[2].view(np.int32))
    n_cases[3] += 1
    tree.close()
print("stress OK: resident loop %d, default ICP vs oracle %d, host radius %d, enqueue-only radius %d cases" % tuple(n_cases))
