import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pcl_amd import capi, sharded
from oracle import pcl_oracle as orc
from tests.test_gpu_icp import surface_cloud, rigid
tgt = surface_cloud(60000, 71)
T0 = rigid(1.5, (1, 0, 1), (0.3, 0.1, -0.2))
src = orc.transform(tgt[::2], np.linalg.inv(T0))
tt = capi.Tree(tgt)
p = capi.icp_default_params()
p.ransac_iterations = 0
for max_it, teps, guess in [(7, 0.0, None), (60, 1e-5, rigid(1.0, (1, 0, 1), (0.2, 0.1, -0.1)))]:
    p.max_iterations, p.transformation_epsilon, p.max_correspondence_distance = max_it, teps, 2.0
    rc1, T1, out1, it1, ok1, pairs1 = sharded.icp_align_gpu_sharded(tt, src, p, guess=guess)
    for ce in (1, 2, 3, 5, 8):
        rc2, T2, out2, it2, ok2, pairs2 = sharded.icp_align_gpu_sharded_dev(tt, src, p, guess=guess, check_every=ce)
        print("case", max_it, "check_every", ce, "it", it1, it2, "Teq", np.array_equal(T1, T2), "out maxdiff", np.abs(out1 - out2).max())
    rc3, T3, out3, st3 = capi.icp_align(tt, src, p, guess=guess)
    print("case", max_it, "icp_align it", st3.iterations, "Teq", np.array_equal(T1, T3), "out maxdiff", np.abs(out1 - out3).max(), "rows differing", int((out1 != out3).any(axis=1).sum()))
