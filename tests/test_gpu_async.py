"""pclmi_knn_dev_async: enqueue-only k-NN batches with stream-ordered workspace reuse.

Batches are queued back to back on two streams (the workspace pool hands one workspace from stream to stream
behind an event, never a host wait); every batch must equal the synchronous pclmi_knn_dev result bit for bit and
the oracle on a sample.  Mirrors the reference's batched loop (search/include/pcl/search/search.h:244-262).
"""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from pcl_amd import capi as c
    assert c.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return c


def test_async_batches_on_two_streams_equal_sync(capi):
    import torch
    from oracle import pcl_oracle as orc
    dev = torch.device("cuda", 0)
    pts = util.uniform_cloud(300_000, 11, 0, 100)
    tree = capi.Tree(pts)
    k = 16
    sizes = [70_000, 50, 40_000, 33_000, 1, 120_000, 64, 65]  # both ordering paths, single-wave batches
    rng = np.random.default_rng(12)
    qs = [(rng.random((n, 3), dtype=np.float32) * 100).astype(np.float32) for n in sizes]
    dq = [torch.from_numpy(q).to(dev) for q in qs]
    oi = [torch.full((len(q), k), -7, dtype=torch.int32, device=dev) for q in qs]
    od = [torch.full((len(q), k), -7.0, dtype=torch.float32, device=dev) for q in qs]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    for rep in range(3):  # the pool's workspaces change hands between the streams
        for b, q in enumerate(qs):
            s = streams[(b + rep) % 2]
            tree.knn_dev_async(dq[b].data_ptr(), len(q), 12, k, oi[b].data_ptr(), od[b].data_ptr(), s.cuda_stream)
    for s in streams:
        s.synchronize()
    for b, q in enumerate(qs):
        si, sd = tree.knn(q, k)
        assert np.array_equal(oi[b].cpu().numpy(), si), "batch %d indices differ from pclmi_knn" % b
        assert np.array_equal(od[b].cpu().numpy().view(np.int32), sd.view(np.int32))
    ci, cd = orc.Tree(pts).knn(qs[0][:5000], k, nthreads=8)
    assert np.array_equal(oi[0][:5000].cpu().numpy(), ci)
    assert np.array_equal(od[0][:5000].cpu().numpy().view(np.int32), cd.view(np.int32))
    tree.close()


def test_async_argument_errors(capi):
    pts = util.uniform_cloud(1000, 3)
    tree = capi.Tree(pts)
    with pytest.raises(capi.PclmiError):
        tree.knn_dev_async(0, 10, 12, 4, 0, 0, None)  # NULL pointers
    with pytest.raises(capi.PclmiError):
        tree.knn_dev_async(8, 10, 10, 4, 8, 8, None)  # bad stride
    tree.knn_dev_async(0, 0, 12, 4, 0, 0, None)  # empty batch: nothing to do
    tree.close()


def test_pipelined_batches_equal_sync(capi):
    """pclmi_knn_dev_pipelined: the ordering pass of call i + 1 runs on the library's side stream under call i's
    traversal; every batch equals pclmi_knn bit for bit, whatever the mix of sizes (alternating workspaces, batches
    without an ordering pass, another entry point in between on the same stream)."""
    import torch
    dev = torch.device("cuda", 0)
    pts = util.uniform_cloud(400_000, 21, 0, 100)
    tree = capi.Tree(pts)
    k = 16
    sizes = [200_000, 150_000, 40, 90_000, 200_000, 1, 35_000, 120_000, 64, 200_000]
    rng = np.random.default_rng(22)
    qs = [(rng.random((n, 3), dtype=np.float32) * 100).astype(np.float32) for n in sizes]
    dq = [torch.from_numpy(q).to(dev) for q in qs]
    oi = [torch.full((len(q), k), -7, dtype=torch.int32, device=dev) for q in qs]
    od = [torch.full((len(q), k), -7.0, dtype=torch.float32, device=dev) for q in qs]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()  # the queries are valid before the first call: the entry point's contract
    for rep in range(3):
        for b, q in enumerate(qs):
            fn = tree.knn_dev_async if (rep == 1 and b % 4 == 3) else tree.knn_dev_pipelined
            fn(dq[b].data_ptr(), len(q), 12, k, oi[b].data_ptr(), od[b].data_ptr(), stream.cuda_stream)
    stream.synchronize()
    for b, q in enumerate(qs):
        si, sd = tree.knn(q, k)
        assert np.array_equal(oi[b].cpu().numpy(), si), "batch %d indices differ from pclmi_knn" % b
        assert np.array_equal(od[b].cpu().numpy().view(np.int32), sd.view(np.int32))
    tree.close()
