import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from tests import util
from pcl_amd import capi
from oracle import pcl_oracle as orc
src, tgt = util.bun0(), util.bun4()
tt = capi.Tree(tgt)
for ransac in (0, 1000):
    p = capi.icp_default_params(); p.max_iterations=50; p.transformation_epsilon=1e-8; p.max_correspondence_distance=0.05; p.ransac_iterations=ransac
    capi.icp_align(tt, src, p)
    t0=time.perf_counter(); rc,T,out,st = capi.icp_align(tt, src, p); dt=time.perf_counter()-t0
    po = orc.icp_default_params(); po.max_iterations=50; po.transformation_epsilon=1e-8; po.max_correspondence_distance=0.05; po.ransac_iterations=ransac
    t0=time.perf_counter(); r2 = orc.icp_align(src, tgt, po); dto=time.perf_counter()-t0
    print("ransac", ransac, "gpu %.2f ms (%d iterations)" % (dt*1e3, st.iterations), "oracle cpu %.2f ms (%d iterations)" % (dto*1e3, r2[3].iterations))
