"""pclmi_comm with TWO processes (one GPU shared: HIP IPC works between processes on one device).

Covers what a one-rank RCCL test cannot: the IPC mapping of the peer mailboxes, the granule protocol over many
rounds, and the sharded device-resident ICP loop whose step kernel exchanges the sums and solves in its last
workgroup - against the single-process pclmi_icp_align on the whole source (1e-5; bit-identical across ranks).
"""
import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_process_allreduce_and_sharded_icp(tmp_path):
    from pcl_amd import capi
    from tests import comm_worker
    assert capi.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    name = "t%s" % uuid.uuid4().hex[:12]
    env = dict(os.environ, PYTHONPATH=str(ROOT), PCLMI_DEVICE_SHARE="2")  # the ranks share one GPU here
    procs = [subprocess.Popen([sys.executable, "-u", str(ROOT / "tests" / "comm_worker.py"), str(r), "2", name,
                               str(tmp_path / ("r%d.npz" % r))], env=env, cwd=str(ROOT),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert int(a["rc"]) == 0 and int(b["rc"]) == 0
    assert np.array_equal(a["T"], b["T"]), "every rank must hold the same transform, bit for bit"
    assert int(a["iterations"]) == int(b["iterations"]) and int(a["pairs"]) == int(b["pairs"])
    # the single-process loop over the whole source
    tgt, src = comm_worker.clouds()
    tree = capi.Tree(tgt)
    p = capi.icp_default_params()
    p.max_iterations, p.ransac_iterations, p.max_correspondence_distance = 12, 0, 2.0
    p.transformation_epsilon = 1e-9
    rc, T, out, st = capi.icp_align(tree, src, p)
    assert rc == 0 and st.iterations == int(a["iterations"]) and st.last_correspondences == int(a["pairs"])
    assert np.abs(T - a["T"]).max() <= 1e-5
    both = np.concatenate([a["aligned"], b["aligned"]])
    assert both.shape == out.shape and np.abs(both[:, :3] - out[:, :3]).max() <= 1e-4
    # empty shard on rank 1: both ranks agree, and equal a single-process run over rank 0's shard
    assert int(a["rc2"]) == 0 and np.array_equal(a["T2"], b["T2"]) and int(a["it2"]) == int(b["it2"]) == 3
    per = (len(src) + 1) // 2
    p.max_iterations = 3
    rc, T3, _, st3 = capi.icp_align(tree, src[:per], p)
    assert st3.last_correspondences == int(a["pairs2"]) and np.abs(T3 - a["T2"]).max() <= 1e-5
    # pclmi_icp_align_sharded: whole source in, Hilbert-ordered contiguous shards, own rows out
    assert int(a["rc3"]) == 0 and np.array_equal(a["T3"], b["T3"]) and int(a["it3"]) == int(b["it3"])
    p.max_iterations = 12
    rc, T, out, st = capi.icp_align(tree, src, p)
    assert int(a["it3"]) == st.iterations and int(a["pairs3"]) == st.last_correspondences
    assert np.abs(a["T3"] - T).max() <= 1e-5
    own_a, own_b = ~np.isnan(a["al3"][:, 0]), ~np.isnan(b["al3"][:, 0])
    assert (own_a ^ own_b).all() and abs(int(own_a.sum()) - int(own_b.sum())) <= 1, "every row belongs to one rank"
    merged = np.where(own_a[:, None], a["al3"], b["al3"])
    assert np.abs(merged[:, :3] - out[:, :3]).max() <= 1e-4
    # point-to-plane estimator over two ranks == one process (32 sums exchanged, 6x6 solve on every rank)
    assert int(a["rc4"]) == 0 and np.array_equal(a["T4"], b["T4"]) and int(a["it4"]) == int(b["it4"])
    rc, T, out, st = capi.icp_align_point_to_plane(tree, src, comm_worker.target_normals(tgt), p)
    assert rc == 0 and int(a["it4"]) == st.iterations and int(a["pairs4"]) == st.last_correspondences
    assert np.abs(a["T4"] - T).max() <= 1e-5
    own_a = ~np.isnan(a["al4"][:, 0])
    merged = np.where(own_a[:, None], a["al4"], b["al4"])
    assert np.abs(merged[:, :3] - out[:, :3]).max() <= 1e-4
    # RANSAC rejection on (the reference's default), sharded == single process: same pairs, same inliers (the same
    # trials were drawn, the same best model chosen), T within 1e-5, bit-identical across the ranks
    tgt_r, src_r = comm_worker.ransac_clouds()
    tree_r = capi.Tree(tgt_r)
    for tag, maxd, plane in (("a", 2.0, False), ("b", 0.25, False), ("c", 2.0, True)):
        pr = capi.icp_default_params()
        pr.max_iterations, pr.ransac_iterations, pr.max_correspondence_distance = 6, 1000, maxd
        pr.ransac_outlier_threshold = 0.1
        if plane:
            rc, T, out, st = capi.icp_align_point_to_plane(tree_r, src_r, comm_worker.target_normals(tgt_r), pr)
        else:
            rc, T, out, st = capi.icp_align(tree_r, src_r, pr)
        assert rc == 0 and int(a["rc5" + tag]) == 0 and int(b["rc5" + tag]) == 0, tag
        assert np.array_equal(a["T5" + tag], b["T5" + tag]), tag
        assert int(a["it5" + tag]) == st.iterations and int(a["pairs5" + tag]) == st.last_correspondences, tag
        assert int(a["inl5" + tag]) == st.last_inliers, (tag, int(a["inl5" + tag]), st.last_inliers)
        assert st.last_inliers < st.last_correspondences, "the rejection step rejects"
        assert st.last_inliers >= 3
        if tag == "b":
            assert st.last_correspondences < len(src_r), "case b must leave rows without a pair"
        assert np.abs(a["T5" + tag] - T).max() <= 1e-5, tag
        own_a = ~np.isnan(a["al5" + tag][:, 0])
        merged = np.where(own_a[:, None], a["al5" + tag], b["al5" + tag])
        assert np.abs(merged[:, :3] - out[:, :3]).max() <= 1e-4, tag
