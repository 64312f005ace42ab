"""Resident ICP loop (development helper; needs make -C pcl_amd/csrc unitvariant UNIT=icp_kernels NAME=lt
EXTRA=-DPCLMI_LOOP_TICKS and PCLMI_LIB=pcl_amd/lib/libpclmi_lt.so): device wall-clock stamps of workgroup 0 at the top
of each of the last 8 iterations of one launch, and the last iteration's arrival / sums / solve stamps.
usage: icp_loop_ticks.py [world ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from pcl_amd import capi

tgt, src_all, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
lib = capi.load_library()
lib.pclmi_debug_icp_walk.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.pclmi_debug_icp_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
for world in [int(v) for v in (sys.argv[1:] or ["8", "64", "1024"])]:
    ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world)
    ctx.run_dev(25.0, 24)
    w = (C.c_uint64 * 8)()
    lib.pclmi_debug_icp_walk(ctx.h, w)
    t = (C.c_uint64 * 4)()
    lib.pclmi_debug_icp_ticks(ctx.h, t)
    tops = sorted(int(v) for v in w)
    per = np.diff(tops) / 100.0
    print("world %d (%d points): iteration periods (us) %s | last iteration: arrival %.1f sums %.1f solve %.1f"
          % (world, ctx.shard_size, np.round(per, 1), (t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0), flush=True)
    ctx.end(want_aligned=False)
