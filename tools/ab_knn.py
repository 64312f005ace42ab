"""A/B helper for kernel variants (development): PCLMI_LIB=<variant .so> python3 tools/ab_knn.py [n] [nq] [k,k,...]
Prints the traversal-kernel time (library HIP events), a checksum of the whole result and an oracle check on a sample."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pcl_amd import capi
if os.environ.get("PCLMI_LIB"):
    capi.load_library(os.environ["PCLMI_LIB"])
from oracle import pcl_oracle as orc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16]
pts = (np.random.default_rng(20260101).random((n, 3), dtype=np.float32) * 1024).astype(np.float32)
q = (np.random.default_rng(20260102).random((nq, 3), dtype=np.float32) * 1024).astype(np.float32)
tree = capi.Tree(pts)
otree = orc.Tree(pts)
dq = torch.from_numpy(q).cuda()
for k in ks:
    di = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for _ in range(3):
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
    torch.cuda.synchronize()
    capi.timing_enable(True)
    capi.timing_read(capi.TIMER_KNN, reset=True)
    for _ in range(10):
        tree.knn_dev(dq.data_ptr(), nq, 12, k, di.data_ptr(), dd.data_ptr())
    torch.cuda.synchronize()
    ms, cnt = capi.timing_read(capi.TIMER_KNN, reset=True)
    capi.timing_enable(False)
    hi, hd = di.cpu().numpy(), dd.cpu().numpy()
    h = hashlib.md5(hi.tobytes() + hd.tobytes()).hexdigest()[:12]
    oi, od = otree.knn(q[:20000], k, nthreads=16)
    ok = np.array_equal(hi[:20000], oi) and np.array_equal(hd[:20000].view(np.int32), od.view(np.int32))
    print("%s k=%d: kernel %.4f ms  md5 %s  oracle(20000) %s" % (os.environ.get("PCLMI_LIB", "default").split("/")[-1],
                                                                 k, ms / max(cnt, 1), h, "OK" if ok else "MISMATCH"), flush=True)
