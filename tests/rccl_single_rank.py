"""Helper of tests/test_gpu_icp.py::test_rccl_all_reduce_between_the_library_kernels (run as a script, one process):
a process group of ONE rank over the nccl (= RCCL) backend, the device-resident sharded ICP driver with the
all-reduce forced on, compared with pclmi_icp_align.  What it proves on a one-GPU box: RCCL initialises in this
image, and torch's collective is ordered correctly between kernels the library launches on torch's raw stream handle
(the N > 1 loop of bench.py and pcl_amd/sharded.py); the multi-rank arithmetic is covered by the gloo test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import bench
from pcl_amd import capi, sharded

port = int(sys.argv[1]) if len(sys.argv) > 1 else 29655
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0,
                        device_id=torch.device("cuda", 0))
tgt = bench.surface_cloud(200_000, 3)
rng = np.random.default_rng(4)
Ti = np.linalg.inv(bench.rigid(1.0, (1, 1, 1), (0.5, -0.3, 0.2)))
src = tgt[rng.permutation(len(tgt))[:60_000]] + rng.normal(0, 0.05, (60_000, 3)).astype(np.float32)
src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
tree = capi.Tree(tgt)
p = capi.icp_default_params()
p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 12, 2.0, 0
rc, T, out, st = capi.icp_align(tree, src, p)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    rc2, T2, out2, it, ok, pairs = sharded.icp_align_gpu_sharded_dev(tree, src, p, force_collective=True)
    # bench.py's loop shape too: raw stepping, one all-reduce per iteration, state read once at the end
    ctx = capi.IcpShard(tree, src)
    sums = torch.zeros(capi.ICP_NSUMS, dtype=torch.float64, device="cuda")
    for _ in range(st.iterations):
        ctx.step_dev(p.max_correspondence_distance, sums.data_ptr(), stream.cuda_stream)
        dist.all_reduce(sums)
        ctx.solve_dev(sums.data_ptr(), stream.cuda_stream)
    rc3, T3, st3 = ctx.state()
    ctx.end(want_aligned=False)
dist.barrier()
dist.destroy_process_group()
assert rc == rc2 == rc3 == 0 and ok and it == st.iterations == st3.iterations and pairs == st.last_correspondences
assert np.array_equal(T, T2) and np.array_equal(out, out2) and np.array_equal(T, T3)
print("RCCL single-rank loop OK: %d iterations, %d pairs" % (it, pairs))
