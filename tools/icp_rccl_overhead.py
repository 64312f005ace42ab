"""Cost of torch's RCCL all-reduce inside the ICP loop (development helper): one-rank nccl group on ONE GPU, bench.py's
N > 1 loop shape (step -> all-reduce -> solve) with and without the collective, rank-0 shard of an N-way split."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist
import bench
from pcl_amd import capi

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29656", world_size=1, rank=0, device_id=torch.device("cuda", 0))
n, iters = 1_000_000, 30
tgt = bench.surface_cloud(n, bench.SEED_T + 10)
T0 = bench.rigid(1.0, (1, 1, 1), (2.0, -1.0, 0.5))
c = np.array([512.0, 512.0, 512.0])
T0[:3, 3] += c - T0[:3, :3] @ c
rng = np.random.default_rng(bench.SEED_Q + 10)
src_all = tgt[rng.permutation(n)] + rng.normal(0, 0.1, (n, 3)).astype(np.float32)
Ti = np.linalg.inv(T0)
src_all = (src_all.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
tree = capi.Tree(tgt)
stream = torch.cuda.Stream()
sums = torch.zeros(capi.ICP_NSUMS, dtype=torch.float64, device="cuda")
with torch.cuda.stream(stream):
    for world in (1, 8):
        src = src_all[:(n + world - 1) // world]
        for coll in (False, True):
            best = 1e9
            for rep in range(3):
                ctx = capi.IcpShard(tree, src)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    ctx.step_dev(25.0, sums.data_ptr(), stream.cuda_stream)
                    if coll:
                        dist.all_reduce(sums)
                    ctx.solve_dev(sums.data_ptr(), stream.cuda_stream)
                t1 = time.perf_counter()
                rc, T, st = ctx.state()
                best = min(best, time.perf_counter() - t0)
                enq = t1 - t0
                ctx.end(want_aligned=False)
            print("shard 1/%d, all-reduce %s: %.4f ms / iteration (host enqueue %.4f ms / iteration)" %
                  (world, "on " if coll else "off", best / iters * 1e3, enq / iters * 1e3), flush=True)
dist.destroy_process_group()
