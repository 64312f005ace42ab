"""Per-iteration step-kernel time of the first iterations of configs[3]'s align from the identity (development helper;
PCLMI_NN1=union switches the step to the wave-uniform walk)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from pcl_amd import capi
tgt, src_all, T0 = bench.icp_clouds()
tree = capi.Tree(tgt)
for rep in range(2):
    capi.timing_enable(2 << capi.TIMER_ICP_STEP)
    ctx = capi.IcpShard(tree, src_all)
    per = []
    for _ in range(12):
        capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        ctx.run_dev(25.0, 1)
        ctx.state()
        ms, cnt = capi.timing_read(capi.TIMER_ICP_STEP, reset=True)
        per.append(ms / max(cnt, 1))
    ctx.end(want_aligned=False)
    capi.timing_enable(False)
print("NN1=%s step kernel us per iteration:" % os.environ.get("PCLMI_NN1", "lane"), " ".join("%.0f" % (v * 1e3) for v in per))
