import ctypes as C, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np, bench
from pcl_amd import capi
tgt, src_all, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
lib = capi.load_library()
lib.pclmi_debug_icp_walk.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.pclmi_debug_icp_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
ctx = capi.IcpShard(tree, src_all, rank=4, world=8)
for it in range(20):
    ctx.run_dev(25.0, 1)
    w = (C.c_uint64 * 8)(); lib.pclmi_debug_icp_walk(ctx.h, w)
    t = (C.c_uint64 * 4)(); lib.pclmi_debug_icp_ticks(ctx.h, t)
    if it >= 15: print("sums->solve begin %.2f  solve_from_sums %.2f  rest %.2f" % ((w[0]-t[2])/100, (w[1]-w[0])/100, (t[3]-w[1])/100))
