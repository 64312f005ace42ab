"""Randomised differential stress of the registration-side entry points against the oracle (development helper, run on
the GPU box: STRESS_SECONDS=120 python tools/stress_registration.py): every rejector on random correspondence lists
(sizes around the block / wavefront boundaries, tie blocks, invalid matches), transform, the SVD estimator on index
lists, fitness score, k-based and radius-based normals, and short point-to-point / point-to-plane aligns with RANSAC
off (bit-for-bit decisions, transform within a few ulps of the extent)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from pcl_amd import capi
from pcl_amd.capi import CORR_DTYPE
from oracle import pcl_oracle as orc
import bench

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
t_end = time.time() + float(os.environ.get("STRESS_SECONDS", "90"))
cases = {"rejectors": 0, "estimators": 0, "normals": 0, "icp": 0, "p2plane": 0}


def fail(msg, **arrs):
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/stress_registration_fail.npz", **arrs)
    print("MISMATCH", msg)
    sys.exit(1)


def random_corr(n):
    c = np.zeros(n, CORR_DTYPE)
    c["index_query"] = rng.permutation(n).astype(np.int32)
    c["index_match"] = rng.integers(-2 if rng.random() < 0.3 else 0, max(1, int(n * rng.uniform(0.05, 2.0))), n).astype(np.int32)
    c["distance"] = (rng.random(n, dtype=np.float32) ** 2).astype(np.float32)
    if n >= 4 and rng.random() < 0.6:
        c["distance"][rng.integers(0, n, max(1, n // int(rng.choice([2, 10, 50]))))] = np.float32(rng.choice([0.0, 0.25, 1.0]))
    return c


while time.time() < t_end:
    # ---- rejectors
    n = int(rng.choice([0, 1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 4097, 30000, 130000]))
    c = random_corr(n)
    md = float(rng.choice([0.1, 0.5, 0.9, 2.0]))
    if not np.array_equal(capi.reject_distance(c, md), orc.reject_distance(c, md)):
        fail("reject_distance n %d md %g" % (n, md), c=c)
    fac = float(rng.choice([0.5, 1.0, 2.0]))
    g, gm = capi.reject_median(c, fac)
    o, om = orc.reject_median(c, fac)
    if n and not (gm == om and np.array_equal(g, o)):
        fail("reject_median n %d factor %g" % (n, fac), c=c)
    if not np.array_equal(capi.reject_one_to_one(c), orc.reject_one_to_one(c)):
        fail("reject_one_to_one n %d" % n, c=c)
    ov, mc = float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])), int(rng.choice([0, 0, 5, 100]))
    if not np.array_equal(capi.reject_trimmed(c, ov, mc), orc.reject_trimmed(c, ov, mc)):
        fail("reject_trimmed n %d overlap %g min %d" % (n, ov, mc), c=c)
    if n:
        a, b, lam = ((0.05, 0.95, 0.95), (0.2, 0.6, 0.5), (0.0, 1.0, 2.0))[int(rng.integers(3))]
        g, gf, gd = capi.reject_var_trimmed(c, a, b, lam, True)
        o, of, od = orc.reject_var_trimmed(c, a, b, lam, True)
        ties = int((c["distance"] == np.float32(gd)).sum()) > 1  # parity unpinned for ties AT the trim distance
        if (gf, gd) != (of, od) or len(g) != len(o) or (not ties and not np.array_equal(g, o)):
            fail("reject_var_trimmed n %d (%g %g %g)" % (n, a, b, lam), c=c)
    cases["rejectors"] += 1
    # ---- clouds
    nt = int(rng.choice([5, 60, 400, 3000, 20000])) + int(rng.integers(0, 7))
    tgt = bench.surface_cloud(max(nt, 50), int(rng.integers(1 << 30)))[:nt]
    variant = rng.random()
    if variant < 0.25:  # twins and triplets: the walk's tie rule (lowest index) against the certificates' 
        tgt = np.concatenate([tgt, tgt[rng.integers(0, nt, nt // 2)], tgt[rng.integers(0, nt, nt // 5)]])[rng.permutation(nt + nt // 2 + nt // 5)]
        nt = len(tgt)
    elif variant < 0.4:  # a lattice: equal distances everywhere
        tgt = (rng.integers(0, 12, (nt, 3)) * 4).astype(np.float32)
    ext = float(np.abs(tgt).max())
    T0 = bench.rigid(float(rng.uniform(0.2, 3.0)), rng.normal(size=3), rng.normal(size=3) * 0.01 * ext)
    sel = rng.permutation(nt)[: max(3, int(nt * rng.uniform(0.3, 1.0)))]
    src = orc.transform(tgt[sel] + rng.normal(0, 2e-4 * ext, (len(sel), 3)).astype(np.float32), np.linalg.inv(T0))
    tree = capi.Tree(tgt)
    # ---- transform, estimator on index lists, fitness
    if not np.array_equal(capi.transform(src, T0).view(np.int32), orc.transform(src, T0).view(np.int32)):
        fail("transform n %d" % len(src), src=src, T=T0)
    m = int(rng.integers(3, len(src) + 1))
    si = rng.integers(0, len(src), m).astype(np.int32)
    ti = sel[si].astype(np.int32)
    Tg = capi.estimate_rigid_svd(src, tgt, si, ti)
    To = orc.estimate_rigid_svd(src, tgt, si, ti, scalar_double=True)
    # (a near-degenerate sample has no unique answer: compare where the pairs span space)
    if np.linalg.matrix_rank(src[si].astype(np.float64) - src[si].mean(0), tol=1e-3 * ext) == 3:
        if np.abs(Tg - To).max() > 2e-5 + 2e-6 * ext:
            fail("estimate_rigid_svd m %d ext %g dT %g" % (m, ext, np.abs(Tg - To).max()), src=src, tgt=tgt, si=si, ti=ti)
    mr = float(rng.choice([0.01 * ext, 1e30]))
    fg, fo = capi.fitness_score(tree, src, T0, mr), orc.fitness_score(src, tgt, T0, mr)
    if not (fg == fo or abs(fg - fo) <= 1e-12 * max(1.0, abs(fo))):
        fail("fitness_score %r / %r" % (fg, fo), src=src, tgt=tgt, T=T0)
    cases["estimators"] += 1
    # ---- normals
    # (not on the lattice: a neighbourhood with a REPEATED smallest eigenvalue has no normal; the reference's closed-form
    # eigenvector - cross products of the rows of A - lambda I - comes out 0/0 = NaN there, the library returns one of
    # the valid directions: parity unpinned)
    if nt >= 20 and not (0.25 <= variant < 0.4):
        otree = orc.Tree(tgt)
        qn = tgt[rng.integers(0, nt, min(nt, 2000))]
        k = int(rng.choice([3, 5, 10, 16, 20, 32, 40]))
        vp = tuple(float(v) for v in rng.normal(size=3) * ext)
        g = capi.normals(tree, qn, k=k, viewpoint=vp)
        o = orc.normals(otree, qn, k=k, viewpoint=vp, nthreads=8)
        ok = np.isfinite(o).all(axis=1) & np.isfinite(g).all(axis=1)
        if variant >= 0.4 and not np.array_equal(np.isfinite(o).all(axis=1), np.isfinite(g).all(axis=1)):  # (twins: degenerate too)
            fail("normals finite rows k %d" % k, tgt=tgt, q=qn, k=k)
        # the direction: against numpy's fp64 eigenvector of the same neighbours (the oracle restates the reference's
        # fp32 one-pass covariance, whose cancellation at coordinates ~1e3 moves well-conditioned normals by 1e-3 and
        # more - the library accumulates in fp64 about the query point), where the two smallest eigenvalues are apart;
        # against the oracle only loosely, on small-extent clouds
        nb = tree.knn(qn, k)[0]
        for i in rng.integers(0, len(qn), 60):
            if not ok[i]:
                continue
            w, v = np.linalg.eigh(np.cov(tgt[nb[i]].astype(np.float64).T, bias=True))
            if (w[1] - w[0]) > 1e-3 * w[2]:
                if abs(abs(float(g[i, :3].astype(np.float64) @ v[:, 0])) - 1) > 1e-6 or abs(g[i, 3] - w[0] / w.sum()) > 1e-5:
                    fail("normals k %d row %d" % (k, i), tgt=tgt, q=qn, k=k)
                # the sign: towards the viewpoint (the oracle's fp32 plane can differ in sign only on a grazing view)
                view = np.asarray(vp, np.float64) - qn[i]
                c = float(view @ g[i, :3])
                if c < -1e-3 * np.linalg.norm(view):
                    fail("normals orientation k %d row %d" % (k, i), tgt=tgt, q=qn, k=k)
        cases["normals"] += 1
    # ---- short aligns, RANSAC off
    if len(src) >= 10:
        p, po = capi.icp_default_params(), orc.icp_default_params()
        mcd = float(rng.choice([0.02, 0.1, 10.0])) * ext
        iters = int(rng.integers(1, 8))
        for x in (p, po):
            x.max_iterations = iters
            x.max_correspondence_distance = mcd
            x.ransac_outlier_threshold = 0.0
            x.ransac_iterations = 0
        po.svd_scalar_double = 1
        rc, T, out, st = capi.icp_align(tree, src, p)
        rco, To, outo, sto = orc.icp_align(src, tgt, po, nthreads=8)
        if (rc == 0) != (rco == 0):
            fail("icp status %d / %d" % (rc, rco), src=src, tgt=tgt, mcd=mcd)
        # (a handful of pairs: rank-deficient sums, no unique rotation - the two sides part ways, legitimately)
        few = min(st.last_correspondences, sto.last_correspondences) < 10
        gi0, gd0 = tree.knn(src, 1)
        k0 = gd0[:, 0].astype(np.float64) < mcd * mcd
        if k0.sum() >= 10:
            S0, T0p = src[k0].astype(np.float64), tgt[gi0[k0, 0]].astype(np.float64)
            sv = np.linalg.svd((S0 - S0.mean(0)).T @ (T0p - T0p.mean(0)))[1]
            few = few or sv[1] < 1e-4 * sv[0]  # (the rotation needs two directions)
        else:
            few = True
        if rc == 0 and not few and (st.iterations != sto.iterations or st.last_correspondences != sto.last_correspondences or
                                    st.converged != sto.converged or np.abs(T - To).max() > 2e-5 + 2e-6 * ext):
            fail("icp n %d mcd %g: it %d/%d pairs %d/%d conv %d/%d dT %g (ext %g)" % (
                len(src), mcd, st.iterations, sto.iterations, st.last_correspondences, sto.last_correspondences,
                st.converged, sto.converged, np.abs(T - To).max(), ext), src=src, tgt=tgt, mcd=mcd)
        cases["icp"] += 1
    # ---- point-to-plane aligns (normals of the target from the library; the same array on both sides)
    if nt >= 400 and len(src) >= 50 and rng.random() < 0.5:
        tn = np.ascontiguousarray(capi.normals(tree, tgt, k=10)[:, :3])
        p, po = capi.icp_default_params(), orc.icp_default_params()
        mcd, iters = float(rng.choice([0.05, 10.0])) * ext, int(rng.integers(1, 6))
        for x in (p, po):
            x.max_iterations = iters
            x.max_correspondence_distance = mcd
            x.ransac_outlier_threshold = 0.0
            x.ransac_iterations = 0
        rc, T, out, st = capi.icp_align_point_to_plane(tree, src, tn, p)
        rco, To, outo, sto = orc.icp_align_point_to_plane(src, tgt, tn, po, nthreads=8)
        if (rc == 0) != (rco == 0):
            fail("p2plane status %d / %d" % (rc, rco), src=src, tgt=tgt, mcd=mcd, tn=tn)
        if rc == 0 and (st.iterations != sto.iterations or st.last_correspondences != sto.last_correspondences or
                        np.abs(T - To).max() > 1e-4 + 1e-5 * ext):
            fail("p2plane n %d mcd %g: it %d/%d pairs %d/%d dT %g (ext %g)" % (
                len(src), mcd, st.iterations, sto.iterations, st.last_correspondences, sto.last_correspondences,
                np.abs(T - To).max(), ext), src=src, tgt=tgt, mcd=mcd, tn=tn)
        cases["p2plane"] += 1
    tree.close()
print("stress_registration OK:", cases)
