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
    vals = rng.standard_normal((world, 12, 32))  # [rank][round][lane]: every rank knows every rank's inputs
    with torch.cuda.stream(stream):
        for rnd in range(12):
            cnt = 1 + (rnd * 5) % 32
            buf = torch.from_numpy(vals[rank, rnd].copy()).to(dev)
            comm.allreduce_f64(buf.data_ptr(), cnt, stream.cuda_stream)
            res.append((cnt, buf))
    stream.synchronize()
    comm.status()
    for rnd, (cnt, buf) in enumerate(res):
        got = buf.cpu().numpy()
        want = np.zeros(32)
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
    np.savez(out, rc4=rc4, T4=T4, al4=al4, it4=st4.iterations, pairs4=st4.last_correspondences, rc3=rc3, T3=T3, al3=al3, it3=st3.iterations, pairs3=st3.last_correspondences, rc=rc, T=T, iterations=st.iterations, converged=st.converged, pairs=st.last_correspondences,
             aligned=aligned, rc2=rc2, T2=T2, it2=st2.iterations, pairs2=st2.last_correspondences)
    comm.close()


if __name__ == "__main__":
    main()
