"""The N > 1 ICP path on CPU: world_size 2 over gloo.  Each rank owns half of the source cloud; the
per-shard moments come from numpy + the CPU oracle's 1-NN here (the GPU kernel is covered by
tests/test_gpu_icp.py::test_sharded_steps_equal_single_shard); what is under test is the driver
(pcl_amd/sharded.py): one all-reduce of 24 fp64 sums per iteration, the shared host solve
(pclmi_icp_solve), identical transforms on every rank, and agreement with single-process ICP."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    from oracle import pcl_oracle as orc
    from pcl_amd import capi, sharded
    from tests import util
    from tests.test_abi_cpu import numpy_sums
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, tgt = util.bun0(), util.bun4()
        per = (len(src) + world - 1) // world
        shard = src[rank * per:(rank + 1) * per].copy()
        tree = orc.Tree(tgt)
        p = capi.icp_default_params()
        p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 30, 0.05, 0
        thr2 = p.max_correspondence_distance ** 2
        state = {"pts": shard, "prev": np.zeros(len(shard), np.float32)}

        def step():
            idx, d2 = tree.knn(state["pts"], 1)
            keep = d2[:, 0].astype(np.float64) < thr2
            sums = numpy_sums(state["pts"][keep], tgt[idx[keep, 0]])
            cur = np.minimum(d2[:, 0], np.float32(thr2))
            sums[16] = cur.astype(np.float64).sum()
            sums[17] = (cur.astype(np.float64) - state["prev"]).sum()
            sums[18] = len(shard)
            state["prev"] = cur
            return sums

        def set_transform(T):
            state["pts"] = orc.transform(state["pts"], T)

        rc, T, it, ok, pairs = sharded.icp_align_sharded(step, set_transform, p)
        q.put((rank, rc, T, it, ok, pairs, state["pts"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_icp_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import pcl_oracle as orc
    from tests import util
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rc0, T0, it0, ok0, pairs0, pts0), (_, rc1, T1, it1, ok1, pairs1, pts1) = res
    assert rc0 == rc1 == 0 and ok0 and ok1 and it0 == it1 == 30 and pairs0 == pairs1
    assert np.array_equal(T0, T1), "ranks must hold bit-identical transforms without a broadcast"
    # same answer as the single-process reference loop, and as the reference's golden matrix
    src, tgt = util.bun0(), util.bun4()
    po = orc.icp_default_params()
    po.max_iterations, po.max_correspondence_distance, po.ransac_iterations, po.svd_scalar_double = 30, 0.05, 0, 1
    rco, To, outo, sto = orc.icp_align(src, tgt, po)
    assert pairs0 == sto.last_correspondences
    assert np.abs(T0 - To).max() <= 1e-5
    assert np.allclose(np.concatenate([pts0, pts1]), outo[:, :3], atol=2e-5)
    g = util.golden("icp_bun0_bun4.json")
    assert np.all(np.abs(T0[:3] - np.array(g["matrix"])[:3]) <= np.array(g["tol"])[:3])


def test_not_enough_correspondences_is_reported():
    from pcl_amd import capi, sharded
    p = capi.icp_default_params()
    rc, T, it, ok, pairs = sharded.icp_align_sharded(lambda: np.zeros(capi.ICP_NSUMS), lambda T: None, p)
    assert rc == 6 and not ok and it == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))
