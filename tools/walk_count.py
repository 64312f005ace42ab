"""Rounds of the per-lane walk per wavefront in converged ICP iterations (development helper; needs
make -C pcl_amd/csrc unitvariant UNIT=icp_kernels NAME=wc EXTRA=-DPCLMI_WALK_COUNT, PCLMI_LIB=pcl_amd/lib/libpclmi_wc.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from pcl_amd import capi
tgt, src_all, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
lib = capi.load_library()
out = (C.c_uint * 4)()
for world in (8, 64):
    ctx = capi.IcpShard(tree, src_all, rank=world // 2, world=world)
    for it in range(16):
        ctx.run_dev(25.0, 1); ctx.state()
        lib.pclmi_debug_walk_count(out)
        if it in (0, 1, 2, 5, 10, 15):
            print("world %d iteration %d: max rounds %d, max descend steps %d, mean rounds %.2f over %d waves" % (world, it + 1, out[0], out[1], out[2] / max(out[3], 1), out[3]), flush=True)
    ctx.end(want_aligned=False)
