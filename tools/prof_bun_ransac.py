"""Default-parameter ICP on the bun fixtures, a few times, for rocprofv3 --kernel-trace (development helper)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from tests import util
from pcl_amd import capi
src, tgt = util.bun0(), util.bun4()
tt = capi.Tree(tgt)
p = capi.icp_default_params(); p.max_iterations = 50; p.transformation_epsilon = 1e-8; p.max_correspondence_distance = 0.05
for _ in range(5):
    t0 = time.perf_counter(); rc, T, out, st = capi.icp_align(tt, src, p); print((time.perf_counter() - t0) * 1e3, st.iterations)
