"""k = 16, 1M x 1M: consecutive batches on ONE stream (pclmi_knn_dev_pipelined) against batches alternating between
TWO streams with their own output buffers (pclmi_knn_dev_async): the second overlaps the tail of a traversal with the
start of the next (development helper)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pcl_amd import capi
n = nq = 1_000_000; k = 16
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
dq = torch.from_numpy(q).cuda()
outs = [(torch.empty((nq, k), dtype=torch.int32, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
steps = 60
def run(mode, ns):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            j = i % ns
            s = streams[j].cuda_stream
            if mode == "pipelined":
                tree.knn_dev_pipelined(dq.data_ptr(), nq, 12, k, outs[j][0].data_ptr(), outs[j][1].data_ptr(), s)
            else:
                tree.knn_dev_async(dq.data_ptr(), nq, 12, k, outs[j][0].data_ptr(), outs[j][1].data_ptr(), s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s, %d stream(s): %.4f ms per batch, %.3e queries/s" % (mode, ns, dt / steps * 1e3, nq * steps / dt), flush=True)
run("pipelined", 1); run("async", 1); run("async", 2); run("async", 3); run("pipelined", 2)
a = outs[0][0].clone(); 
tree.knn_dev(dq.data_ptr(), nq, 12, k, outs[1][0].data_ptr(), outs[1][1].data_ptr())
torch.cuda.synchronize(); print("same results:", bool((a == outs[1][0]).all()))
