"""Worker of tests/test_gpu_icp.py::test_certificates_change_nothing: a fixed set of aligns, results to an .npz.
Run in a process of its own because PCLMI_ICP_CERT is read once per process."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def cases():
    import bench as util  # surface_cloud, rigid
    rng = np.random.default_rng(77)
    out = []
    for kind in ("surface", "twins", "lattice", "dense_noise"):
        n = 30_000
        if kind == "lattice":
            tgt = (rng.integers(0, 24, (n, 3)) * 2).astype(np.float32)
        else:
            tgt = util.surface_cloud(n, 78)
        if kind == "twins":
            tgt = np.concatenate([tgt, tgt[rng.integers(0, n, n // 2)], tgt[rng.integers(0, n, n // 4)]])
            tgt = tgt[rng.permutation(len(tgt))]
        noise = {"surface": 0.02, "twins": 0.02, "lattice": 0.3, "dense_noise": 2.0}[kind]  # against a spacing of ~2-8
        T0 = util.rigid(1.5, (1, 2, 3), (0.8, -0.5, 0.3))
        sel = rng.permutation(len(tgt))[: len(tgt) * 3 // 4]
        src = tgt[sel] + rng.normal(0, noise, (len(sel), 3)).astype(np.float32)
        Ti = np.linalg.inv(T0)
        src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        out.append((kind, tgt, src))
    return out


def run(capi):
    res = {}
    for kind, tgt, src in cases():
        tree = capi.Tree(tgt)
        for ransac in (0, 1):
            p = capi.icp_default_params()
            p.max_iterations, p.max_correspondence_distance = 12, 6.0
            p.ransac_iterations, p.ransac_outlier_threshold = (1000, 1.0) if ransac else (0, 0.0)
            rc, T, out, st = capi.icp_align(tree, src, p)
            res["%s_%d_T" % (kind, ransac)] = T
            res["%s_%d_out" % (kind, ransac)] = out
            res["%s_%d_stats" % (kind, ransac)] = np.array([rc, st.iterations, st.converged, st.last_correspondences,
                                                            st.last_inliers], np.int64)
        # the sharded context, iteration by iteration and as one resident launch
        for mode in ("one", "loop"):
            ctx = capi.IcpShard(tree, src)
            if mode == "one":
                for _ in range(8):
                    ctx.run_dev(6.0, 1)
            else:
                ctx.run_dev(6.0, 8)
            rc, T, st = ctx.state()
            res["%s_shard_%s_T" % (kind, mode)] = T
            res["%s_shard_%s_out" % (kind, mode)] = ctx.end()
        tn = np.ascontiguousarray(capi.normals(tree, tgt, k=10)[:, :3])
        p = capi.icp_default_params()
        p.max_iterations, p.max_correspondence_distance, p.ransac_iterations = 8, 6.0, 0
        rc, T, out, st = capi.icp_align_point_to_plane(tree, src, tn, p)
        res["%s_plane_T" % kind] = T
        res["%s_plane_out" % kind] = out
        tree.close()
    return res


if __name__ == "__main__":
    from pcl_amd import capi
    np.savez(sys.argv[1], **run(capi))
