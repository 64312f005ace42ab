"""One rank of the 2-process pclmi_comm test (started by tests/test_gpu_comm.py, never collected by pytest).

usage: comm_worker.py <rank> <world> <shm name> <out.npz>
Both ranks share GPU 0 (HIP IPC works between processes on one device): the mailbox protocol, the IPC mapping and the
fused exchange + solve inside the ICP step kernel are exercised for real; only the xGMI hop is missing.
"""
import sys

import numpy as np


def clouds():
    rng = np.random.default_rng(77)
    tgt = (rng.random((60_000, 3), dtype=np.float32) * 100).astype(np.float32)
    tgt[:, 2] = 3 * np.sin(tgt[:, 0] * 0.2) + 2 * np.cos(tgt[:, 1] * 0.15)
    th = np.deg2rad(1.5)
    T0 = np.eye(4)
    T0[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    T0[:3, 3] = [0.3, -0.2, 0.1]
    Ti = np.linalg.inv(T0)
    src = (tgt[::2].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return tgt, src


def target_normals(tgt):
    """Analytic unit normals of the surface clouds() samples."""
    n = np.stack([-0.6 * np.cos(tgt[:, 0] * 0.2), 0.3 * np.sin(tgt[:, 1] * 0.15), np.ones(len(tgt))], axis=1)
    return (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)


def main():
    rank, world, name, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    from pcl_amd import capi
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    comm = capi.Comm.from_shm(rank, world, name)
    comm.set_timeout(20.0)
    assert comm.rank == rank and comm.size == world
    stream = torch.cuda.Stream(device=dev)
    # ---- plain all-reduces: several rounds back to back (both parities, sequence numbers), uneven counts
    res = []
    rng = np.random.default_rng(5)
    vals = rng.standard_normal((world, 12, 64))  # [rank][round][lane]: every rank knows every rank's inputs
    with torch.cuda.stream(stream):
        for rnd in range(12):
            cnt = 1 + (rnd * 11) % 64
            buf = torch.from_numpy(vals[rank, rnd].copy()).to(dev)
            comm.allreduce_f64(buf.data_ptr(), cnt, stream.cuda_stream)
            res.append((cnt, buf))
    stream.synchronize()
    comm.status()
    for rnd, (cnt, buf) in enumerate(res):
        got = buf.cpu().numpy()
        want = np.zeros(64)
        for r in range(world):  # rank order, like the device code: bit-identical
            want = want + vals[r, rnd]
        assert np.array_equal(got[:cnt], want[:cnt]), (rnd, got[:cnt], want[:cnt])
        assert np.array_equal(got[cnt:], vals[rank, rnd, cnt:]), "lanes past count must stay untouched"
    # ---- sharded ICP, device-resident: ONE kernel per iteration per rank, exchange + solve in its last workgroup
    tgt, src = clouds()
    tree = capi.Tree(tgt)
    per = (len(src) + world - 1) // world
    shard = src[rank * per:(rank + 1) * per]
    p = capi.icp_default_params()
    p.max_iterations, p.ransac_iterations, p.max_correspondence_distance = 12, 0, 2.0
    p.transformation_epsilon = 1e-9
    ctx = capi.IcpShard(tree, shard)
    ctx.set_criteria(p)
    ctx.set_comm(comm)
    ctx.run_dev(p.max_correspondence_distance, p.max_iterations, stream.cuda_stream)
    rc, T, st = ctx.state()
    aligned = ctx.end()
    # an empty shard on one rank: the exchange still runs, the sums are the other rank's
    empty = np.zeros((0, 3), np.float32)
    ctx2 = capi.IcpShard(tree, shard if rank == 0 else empty)
    ctx2.set_criteria(p)
    ctx2.set_comm(comm)
    ctx2.run_dev(p.max_correspondence_distance, 3, stream.cuda_stream)
    rc2, T2, st2 = ctx2.state()
    ctx2.end(want_aligned=False)
    comm.status()
    # the one-call form: every rank passes the whole source, the library cuts the Hilbert-ordered run of this rank
    rc3, T3, al3, st3 = capi.icp_align_sharded(tree, src, comm, p)
    # the same with TransformationEstimationPointToPlaneLLS: 32 sums per exchange, 6x6 solve in the last workgroup
    rc4, T4, al4, st4 = capi.icp_align_point_to_plane(tree, src, target_normals(tgt), p, comm=comm)
    # ---- the reference's DEFAULT loop over two ranks: RANSAC rejection on (icp.hpp:113-150)
    # (a) every row has a pair, 70 % inliers: the first batch of trials does not settle the stopping rule (host-fed
    #     batches), later iterations do;  (b) a tight distance threshold: some rows have no pair (pair mask exchanged
    #     through the bulk window, rank / select);  (c) the point-to-plane estimator's refit
    tgt_r, src_r = ransac_clouds()
    tree_r = capi.Tree(tgt_r)
    res = {}
    for tag, maxd, plane in (("a", 2.0, False), ("b", 0.25, False), ("c", 2.0, True)):
        pr = capi.icp_default_params()
        pr.max_iterations, pr.ransac_iterations, pr.max_correspondence_distance = 6, 1000, maxd
        pr.ransac_outlier_threshold = 0.1
        if plane:
            r = capi.icp_align_point_to_plane(tree_r, src_r, target_normals(tgt_r), pr, comm=comm)
        else:
            r = capi.icp_align_sharded(tree_r, src_r, comm, pr)
        res["rc5" + tag], res["T5" + tag], res["al5" + tag] = r[0], r[1], r[2]
        res["it5" + tag], res["pairs5" + tag], res["inl5" + tag] = r[3].iterations, r[3].last_correspondences, r[3].last_inliers
    comm.status()
    np.savez(out, **res, rc4=rc4, T4=T4, al4=al4, it4=st4.iterations, pairs4=st4.last_correspondences, rc3=rc3, T3=T3, al3=al3, it3=st3.iterations, pairs3=st3.last_correspondences, rc=rc, T=T, iterations=st.iterations, converged=st.converged, pairs=st.last_correspondences,
             aligned=aligned, rc2=rc2, T2=T2, it2=st2.iterations, pairs2=st2.last_correspondences)
    # ---- a rank that skips an exchange: the other one times out ONCE, then the communicator is dead and fails fast
    import time
    comm.set_timeout(1.0)
    if rank == 0:
        buf = torch.zeros(8, dtype=torch.float64, device=dev)
        comm.allreduce_f64(buf.data_ptr(), 8, stream.cuda_stream)
        t0 = time.perf_counter()
        try:
            comm.status()
            raise AssertionError("the exchange with a missing peer must fail")
        except capi.PclmiError as e:
            assert e.code == 7, e
        waited = time.perf_counter() - t0
        assert 0.5 < waited < 10.0, waited
        t0 = time.perf_counter()
        for call in (lambda: comm.allreduce_f64(buf.data_ptr(), 8, stream.cuda_stream),
                     lambda: capi.icp_align_sharded(tree, src, comm, p)):
            try:
                r = call()
                assert r is not None and r[0] == 7, "a dead communicator must refuse further work"
            except capi.PclmiError as e:
                assert e.code == 7, e
        assert time.perf_counter() - t0 < 0.9, "a dead communicator fails at once, not after another timeout"
    else:
        time.sleep(4.0)  # stays alive (its mailbox is mapped by rank 0) while rank 0 times out
    comm.close()


def ransac_clouds():
    """clouds() with 30 % of the source points pushed off the surface: pairs under the distance threshold, but
    outliers of the rigid model."""
    tgt, src = clouds()
    rng = np.random.default_rng(78)
    src = src[:20_000].copy()
    bad = rng.random(len(src)) < 0.3
    src[bad] += (rng.random((int(bad.sum()), 3), dtype=np.float32) - 0.5) * np.float32(1.2)
    return tgt, src


if __name__ == "__main__":
    main()
