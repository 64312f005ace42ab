"""BASELINE configs[4] as a workload: one GPU's share (12.5M of the 100M queries, k = 32) streamed in 4M batches
against the 10M-point target (pcl_amd/sweep.py over pclmi_knn_dev_async), results kept in HBM.

12.5M x 32 neighbours cannot go through the CPU oracle in seconds, so the whole output is checked on the GPU through
size-independent properties - (d2, index) order, index range, uniqueness, every distance recomputed bit-exactly from
the coordinates (separate fp32 torch ops: one rounding per operation) - plus a seeded sample against the oracle's
FLANN-style tree and one batch against the synchronous entry point.
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SEED_T, SEED_Q = 20260101, 20260102
N_TARGET, N_TOTAL, WORLD, K, BATCH = 10_000_000, 100_000_000, 8, 32, 4_000_000


def test_c5_one_gpu_share_streamed():
    import torch
    from oracle import pcl_oracle as orc
    from pcl_amd import capi, sweep
    assert capi.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    dev = torch.device("cuda", 0)
    pts = util.uniform_cloud(N_TARGET, SEED_T, 0, 2048)
    lo, hi = sweep.rank_slice(N_TOTAL, 3, WORLD)  # rank 3's share
    assert hi - lo == 12_500_000
    q = util.uniform_cloud(hi - lo, SEED_Q + 3, 0, 2048)
    tree = capi.Tree(pts)
    tm = {}
    idx, d2 = sweep.knn_sweep(tree, q, K, batch=BATCH, device=dev, timings=tm)
    torch.cuda.synchronize()
    n = len(q)
    assert idx.shape == (n, K) and d2.shape == (n, K)
    print("configs[4] share: %d queries in %.3f s (upload-inclusive) = %.3e queries/s" % (n, tm["seconds"], n / tm["seconds"]))
    p_dev = torch.from_numpy(pts).to(dev)
    q_dev = torch.from_numpy(q).to(dev)
    step = 500_000
    for a in range(0, n, step):
        i, d = idx[a:a + step], d2[a:a + step]
        assert int(i.min()) >= 0 and int(i.max()) < N_TARGET
        dd = d[:, 1:] - d[:, :-1]
        assert bool((d[:, 1:] >= d[:, :-1]).all()), "ascending distances"
        ties = dd == 0
        assert bool((i[:, 1:][ties] > i[:, :-1][ties]).all()), "equal distances in ascending index order"
        si = torch.sort(i, dim=1).values
        assert bool((si[:, 1:] > si[:, :-1]).all()), "no neighbour twice"
        nb = p_dev[i.long()]  # [m, K, 3]
        df = q_dev[a:a + step, None, :] - nb
        sq = df * df
        rec = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        assert torch.equal(rec.view(torch.int32), d.view(torch.int32)), "every distance is the fp32 distance of its point"
        del nb, df, sq, rec
    # the same share searched in one call (resident=True): identical rows
    ri, rd = sweep.knn_sweep(tree, q, K, batch=BATCH, device=dev, resident=True, timings=tm)
    torch.cuda.synchronize()
    print("configs[4] share, one search: %.3f s = %.3e queries/s" % (tm["seconds"], n / tm["seconds"]))
    assert torch.equal(ri, idx) and torch.equal(rd.view(torch.int32), d2.view(torch.int32))
    del ri, rd
    # a batch through the synchronous entry point gives the same rows
    m = 300_000
    a = 4_100_000  # straddles the first batch boundary region of the sweep's second batch
    ci = torch.empty((m, K), dtype=torch.int32, device=dev)
    cd = torch.empty((m, K), dtype=torch.float32, device=dev)
    tree.knn_dev(q_dev[a:a + m].data_ptr(), m, 12, K, ci.data_ptr(), cd.data_ptr(), None)
    assert torch.equal(ci, idx[a:a + m]) and torch.equal(cd.view(torch.int32), d2[a:a + m].view(torch.int32))
    # seeded sample against the oracle's FLANN-style tree: bit-exact
    sel = np.sort(np.random.default_rng(9).choice(n, 3_000, replace=False))
    oi, od = orc.Tree(pts).knn(q[sel], K, nthreads=16)
    gi = idx[torch.from_numpy(sel).to(dev)].cpu().numpy()
    gd = d2[torch.from_numpy(sel).to(dev)].cpu().numpy()
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.int32), od.view(np.int32))
    tree.close()


def test_sweep_small_and_ragged():
    import torch
    from pcl_amd import capi, sweep
    dev = torch.device("cuda", 0)
    pts = util.uniform_cloud(50_000, 1, 0, 10)
    tree = capi.Tree(pts)
    for n, batch in ((0, 10), (1, 10), (1000, 333), (70_001, 33_000)):
        q = util.uniform_cloud(n, 2 + n, 0, 10)
        for resident in (False, True):
            idx, d2 = sweep.knn_sweep(tree, q, 5, batch=batch, device=dev, resident=resident)
            torch.cuda.synchronize()
            if n == 0:
                assert idx.shape == (0, 5)
                continue
            ri, rd = tree.knn(q, 5)
            assert np.array_equal(idx.cpu().numpy(), ri)
            assert np.array_equal(d2.cpu().numpy().view(np.int32), rd.view(np.int32))
    assert sweep.rank_slice(10, 3, 4) == (9, 10) and sweep.rank_slice(10, 0, 4) == (0, 3)
    tree.close()
