"""Where the search part of an ICP iteration spends its time (development helper; needs the instrumented library:
make -C pcl_amd/csrc variant NAME=ticks EXTRA=-DPCLMI_ICP_TICKS, then PCLMI_LIB=pcl_amd/lib/libpclmi_ticks.so).
Stamps of wave 0 of the middle workgroup: start, point loaded, start bound known, common root found, nearest found,
results stored (us since the wave started), for shards of bench.py's configs[3].
usage: icp_walk_ticks.py [world ...]"""
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
for world in [int(v) for v in (sys.argv[1:] or ["1", "8", "1024"])]:
    ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world)
    rows, tails = [], []
    for it in range(30):
        ctx.run_dev(25.0, 1)
        w = (C.c_uint64 * 8)()
        lib.pclmi_debug_icp_walk(ctx.h, w)
        t = (C.c_uint64 * 4)()
        lib.pclmi_debug_icp_ticks(ctx.h, t)
        rows.append([(w[i] - w[0]) / 100.0 for i in range(1, 6)] + [(w[6] - t[0]) / 100.0, w[7] / 100.0])
        tails.append([(t[i] - t[0]) / 100.0 for i in (1, 2, 3)])
    ctx.end(want_aligned=False)
    r, tl = np.array(rows), np.array(tails)
    print("world %d (%d points)" % (world, ctx.shard_size))
    for name, sl in (("iteration 1", slice(0, 1)), ("iterations 2-5", slice(1, 5)), ("iterations 11-30", slice(10, 30))):
        print("   %-16s wave: point %.1f  bound %.1f  root %.1f  nearest %.1f  stored %.1f | latest wave start %.1f  longest wave %.1f | launch: arrival %.1f  sums %.1f  solve %.1f"
              % ((name,) + tuple(r[sl].mean(0)) + tuple(tl[sl].mean(0))), flush=True)
