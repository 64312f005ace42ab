"""Streamed k-NN sweep: BASELINE configs[4] (100M queries, k = 32, against a 10M-point target) as a workload.

The batched form of the reference's per-point loop (search/include/pcl/search/search.h:244-262) at a size where the
queries do not arrive in one piece: a rank's queries are uploaded in batches on a copy stream while the previous
batch is searched on the compute stream (pclmi_knn_dev_async: enqueue-only, the workspace changes hands in stream
order); the neighbour lists stay in HBM.  Queries shard over ranks with no collective.  torch is plumbing: pinned host
memory, device buffers, streams and events.

Batch size (round 3).  What a wavefront of the traversal costs is the union of its 64 queries' search balls, so a batch
should be as DENSE as the target: a 4M batch on the 10M-point target is 0.4 queries per target point and costs 45 %
more kernel time per query than 12.5M queries searched together (measured: 18.1 ms in four launches against 12.5 ms in
one).  HBM is 288 GB - a 12.5M-query batch is 200 MB of queries and 3.2 GB of results - so the default batch is
DEFAULT_BATCH = 12.5M queries (one GPU's share of the 100M on 8 GPUs): the upload of batch i + 1 still overlaps the
search of batch i, and every batch is searched at full density.
"""
import numpy as np


def rank_slice(n_total, rank, world):
    """Contiguous share of `n_total` queries of rank `rank` (the last rank takes the remainder)."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


DEFAULT_BATCH = 12_500_000


def knn_sweep(tree, queries, k, batch=DEFAULT_BATCH, device=None, out=None, timings=None, resident=False):
    """queries: (n, 3 | 4) float32 host array (numpy, or a pinned torch tensor).  Returns (idx, d2) CUDA tensors
    [n, k_eff] resident in HBM.  timings (dict, optional) receives the wall time of the pipeline in seconds.

    resident=False: every uploaded batch is searched on its own while the next one uploads (bounded device memory
    for the queries: two batches).  resident=True: the batches are uploaded into one device array and searched in ONE
    call - a GPU's whole share fits HBM many times over (12.5M queries = 150 MB), and one call orders ALL its queries
    along the Hilbert curve, so a wavefront's 64 queries are 3x closer together than in a 4M batch: the traversal
    kernel runs ~1.6x faster per query on the 10M-point target, which outweighs the lost upload overlap."""
    import time

    import torch
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if isinstance(queries, np.ndarray):
        q = torch.from_numpy(np.ascontiguousarray(queries, np.float32))
    else:
        q = queries
    assert q.dim() == 2 and q.shape[1] in (3, 4) and q.dtype == torch.float32
    if not q.is_pinned():
        q = q.pin_memory()  # asynchronous uploads need page-locked memory
    n, w = q.shape
    k_eff = min(k, len(tree))
    if out is None:
        idx = torch.empty((n, k_eff), dtype=torch.int32, device=dev)
        d2 = torch.empty((n, k_eff), dtype=torch.float32, device=dev)
    else:
        idx, d2 = out
        assert idx.shape == (n, k_eff) and d2.shape == (n, k_eff)
    if n == 0 or k_eff == 0:
        return idx, d2
    batch = max(1, min(batch, n))
    if resident:
        return _sweep_resident(tree, q, k, batch, dev, idx, d2, timings)
    stage = [torch.empty((batch, w), dtype=torch.float32, device=dev) for _ in range(2)]
    up, run = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    uploaded = [torch.cuda.Event() for _ in range(2)]
    searched = [torch.cuda.Event() for _ in range(2)]
    cur = torch.cuda.current_stream(dev)
    up.wait_stream(cur)
    run.wait_stream(cur)
    t0 = time.perf_counter()
    for b, a in enumerate(range(0, n, batch)):
        m = min(batch, n - a)
        s = b % 2
        with torch.cuda.stream(up):
            if b >= 2:
                up.wait_event(searched[s])  # the staging buffer is free once its previous batch has been searched
            stage[s][:m].copy_(q[a:a + m], non_blocking=True)
            uploaded[s].record(up)
        run.wait_event(uploaded[s])
        tree.knn_dev_async(stage[s].data_ptr(), m, w * 4, k, idx[a:a + m].data_ptr(), d2[a:a + m].data_ptr(),
                           run.cuda_stream)
        searched[s].record(run)
    run.synchronize()
    if timings is not None:
        timings["seconds"] = time.perf_counter() - t0
    cur.wait_stream(run)
    return idx, d2


def _sweep_resident(tree, q, k, batch, dev, idx, d2, timings):
    import time

    import torch
    n, w = q.shape
    qd = torch.empty((n, w), dtype=torch.float32, device=dev)
    up = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    up.wait_stream(cur)
    t0 = time.perf_counter()
    with torch.cuda.stream(up):
        for a in range(0, n, batch):
            qd[a:a + batch].copy_(q[a:a + batch], non_blocking=True)
        tree.knn_dev_async(qd.data_ptr(), n, w * 4, k, idx.data_ptr(), d2.data_ptr(), up.cuda_stream)
    up.synchronize()
    if timings is not None:
        timings["seconds"] = time.perf_counter() - t0
    cur.wait_stream(up)
    return idx, d2
