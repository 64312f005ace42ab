"""pclmi_knn_dev_noalloc (scratch in a static device array, no hipMalloc) against pclmi_knn_dev.

Two calls in a row through the C ABI, the second with a smaller input (stale scratch contents of the first call
must not leak into it), each bit-exact against the existing entry point; then the size limits.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_noalloc_equals_knn_dev_two_calls_in_a_row():
    import torch
    from pcl_amd import capi
    assert capi.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    lib = capi.load_library()
    dev = torch.device("cuda", 0)
    pts = util.uniform_cloud(400_000, 21, 0, 100)
    tree = capi.Tree(pts)
    stream = torch.cuda.Stream(device=dev)
    for k in (16, 5, 1, 32):
        for nq in (250_000, 20_000, 40):  # two-level order, then the one-level order on a smaller input, one wave
            q = util.uniform_cloud(nq, 22 + nq + k, -5, 105)  # some queries outside the tree's box
            dq = torch.from_numpy(q).to(dev)
            ia = torch.full((nq, k), -3, dtype=torch.int32, device=dev)
            da = torch.full((nq, k), -3.0, dtype=torch.float32, device=dev)
            ib, db = torch.empty_like(ia), torch.empty_like(da)
            torch.cuda.synchronize()
            rc = lib.pclmi_knn_dev_noalloc(tree.h, dq.data_ptr(), nq, 12, k, ia.data_ptr(), da.data_ptr(),
                                           stream.cuda_stream)
            assert rc == 0, lib.pclmi_last_error()
            stream.synchronize()
            rc = lib.pclmi_knn_dev(tree.h, dq.data_ptr(), nq, 12, k, ib.data_ptr(), db.data_ptr(), stream.cuda_stream)
            assert rc == 0, lib.pclmi_last_error()
            assert torch.equal(ia, ib), "indices differ (k=%d, nq=%d)" % (k, nq)
            assert torch.equal(da.view(torch.int32), db.view(torch.int32)), "distances differ (k=%d, nq=%d)" % (k, nq)
    # the oracle agrees too (bit-exact), on the last small batch
    from oracle import pcl_oracle as orc
    oi, od = orc.Tree(pts).knn(q, 32, nthreads=4)
    assert np.array_equal(ia.cpu().numpy(), oi) and np.array_equal(da.cpu().numpy().view(np.int32), od.view(np.int32))
    # the largest batch the static scratch takes (PCLMI_NOALLOC_MAX_QUERIES = 2^20: 131 073 tile counters, the scan's
    # temporary storage checked against its slice of the array), then a small one again
    nq, k = 1 << 20, 8
    q = util.uniform_cloud(nq, 99, 0, 100)
    dq = torch.from_numpy(q).to(dev)
    ia = torch.full((nq, k), -3, dtype=torch.int32, device=dev)
    da = torch.full((nq, k), -3.0, dtype=torch.float32, device=dev)
    ib, db = torch.empty_like(ia), torch.empty_like(da)
    torch.cuda.synchronize()
    rc = lib.pclmi_knn_dev_noalloc(tree.h, dq.data_ptr(), nq, 12, k, ia.data_ptr(), da.data_ptr(), stream.cuda_stream)
    assert rc == 0, lib.pclmi_last_error()
    stream.synchronize()
    rc = lib.pclmi_knn_dev(tree.h, dq.data_ptr(), nq, 12, k, ib.data_ptr(), db.data_ptr(), stream.cuda_stream)
    assert rc == 0, lib.pclmi_last_error()
    assert torch.equal(ia, ib) and torch.equal(da.view(torch.int32), db.view(torch.int32))
    rc = lib.pclmi_knn_dev_noalloc(tree.h, dq.data_ptr(), 3000, 12, k, ia.data_ptr(), da.data_ptr(), stream.cuda_stream)
    assert rc == 0, lib.pclmi_last_error()
    stream.synchronize()
    assert torch.equal(ia[:3000], ib[:3000]) and torch.equal(da[:3000].view(torch.int32), db[:3000].view(torch.int32))
    # limits: a clear error, nothing enqueued
    big = (1 << 20) + 1
    rc = lib.pclmi_knn_dev_noalloc(tree.h, dq.data_ptr(), big, 12, 4, ia.data_ptr(), da.data_ptr(), None)
    assert rc == 5 and b"PCLMI_NOALLOC_MAX_QUERIES" in lib.pclmi_last_error()
    rc = lib.pclmi_knn_dev_noalloc(tree.h, dq.data_ptr(), 40, 12, 33, ia.data_ptr(), da.data_ptr(), None)
    assert rc == 5 and b"exceeds 32" in lib.pclmi_last_error()
    tree.close()
