"""ctypes binding of oracle/liboracle.so - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (pcl_amd) never does.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

c_i64 = C.c_int64
c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)

CORR_DTYPE = np.dtype([("index_query", np.int32), ("index_match", np.int32), ("distance", np.float32)])


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("ransac_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("euclidean_fitness_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("ransac_outlier_threshold", C.c_double),
        ("min_correspondences", C.c_int),
        ("svd_scalar_double", C.c_int),
    ]


class IcpStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("last_correspondences", C.c_int),
        ("fitness_delta", C.c_double),
    ]


def build(force=False):
    so = _HERE / "liboracle.so"
    src = _HERE / "pcl_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_HERE), "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # PCL_ORACLE_LIB: another build of the same source (bench.py's cpu_baseline leg compiles one with
        # -O3 -march=native on the box it runs on: a -march=native library built elsewhere may not run there)
        so = Path(os.environ.get("PCL_ORACLE_LIB") or (_HERE / "liboracle.so"))
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        L.orc_knn_brute.restype = C.c_int
        L.orc_knn_brute.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, C.c_int, c_i32p, c_f32p]
        L.orc_radius_brute.restype = C.c_int
        L.orc_radius_brute.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, C.c_float, C.c_uint32,
                                       C.c_int, c_i64p, C.POINTER(c_i32p), C.POINTER(c_f32p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_tree_build.restype = C.c_void_p
        L.orc_tree_build.argtypes = [c_f32p, c_i64, C.c_int, C.c_int]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_knn.restype = C.c_int
        L.orc_tree_knn.argtypes = [C.c_void_p, c_f32p, c_i64, C.c_int, C.c_int, c_i32p, c_f32p, C.c_int]
        L.orc_tree_radius.restype = C.c_int
        L.orc_tree_radius.argtypes = [C.c_void_p, c_f32p, c_i64, C.c_int, C.c_float, C.c_uint32, C.c_int,
                                      c_i64p, C.POINTER(c_i32p), C.POINTER(c_f32p), C.c_int]
        L.orc_transform.argtypes = [c_f32p, c_i64, C.c_int, c_f32p]
        L.orc_estimate_rigid_svd.argtypes = [c_f32p, C.c_int, c_i32p, c_f32p, C.c_int, c_i32p, c_i64, c_f32p]
        L.orc_estimate_rigid_svd_f64.argtypes = L.orc_estimate_rigid_svd.argtypes
        L.orc_correspondences.restype = c_i64
        L.orc_correspondences.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, C.c_double, C.c_int,
                                          C.c_void_p]
        for name in ("orc_reject_distance",):
            getattr(L, name).restype = c_i64
        L.orc_reject_distance.argtypes = [C.c_void_p, c_i64, C.c_float, C.c_void_p]
        L.orc_reject_surface_normal.restype = c_i64
        L.orc_reject_surface_normal.argtypes = [C.c_void_p, c_i64, c_f32p, C.c_int, c_f32p, C.c_int, C.c_double,
                                                C.c_void_p]
        L.orc_reject_median.restype = c_i64
        L.orc_reject_median.argtypes = [C.c_void_p, c_i64, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_reject_one_to_one.restype = c_i64
        L.orc_reject_one_to_one.argtypes = [C.c_void_p, c_i64, C.c_void_p]
        L.orc_reject_trimmed.restype = c_i64
        L.orc_reject_trimmed.argtypes = [C.c_void_p, c_i64, C.c_float, C.c_uint, C.c_void_p]
        L.orc_reject_var_trimmed.restype = c_i64
        L.orc_reject_var_trimmed.argtypes = [C.c_void_p, c_i64, C.c_double, C.c_double, C.c_double, C.c_int,
                                             C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_ransac_registration.restype = c_i64
        L.orc_ransac_registration.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, c_i32p, c_i32p,
                                              c_i64, C.c_double, C.c_int, c_i32p, c_f32p, C.POINTER(C.c_int)]
        L.orc_icp_default_params.argtypes = [C.POINTER(IcpParams)]
        L.orc_icp_align.restype = C.c_int
        L.orc_icp_align.argtypes = [C.POINTER(IcpParams), c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, c_f32p,
                                    c_f32p, c_f32p, C.POINTER(IcpStats), C.c_int]
        L.orc_icp_align_scaled.restype = C.c_int
        L.orc_icp_align_scaled.argtypes = [C.POINTER(IcpParams), c_f32p] + list(L.orc_icp_align.argtypes[1:])
        L.orc_icp_align_point_to_plane.restype = C.c_int
        L.orc_icp_align_point_to_plane.argtypes = [C.POINTER(IcpParams), c_f32p, c_i64, C.c_int, c_f32p, C.c_int,
                                                   c_f32p, c_i64, C.c_int, c_f32p, C.c_int, c_f32p, c_f32p, c_f32p,
                                                   C.POINTER(IcpStats), C.c_int]
        L.orc_fitness_score.restype = C.c_double
        L.orc_fitness_score.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, c_i64, C.c_int, c_f32p, C.c_double]
        L.orc_estimate_point_to_plane_lls.argtypes = [c_f32p, C.c_int, c_f32p, C.c_int, c_i32p, c_f32p, C.c_int, c_f32p,
                                                      C.c_int, c_i32p, c_i64, c_f32p]
        L.orc_correspondences_normal_shooting.restype = c_i64
        L.orc_correspondences_normal_shooting.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, C.c_int, c_f32p, c_i64,
                                                          C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_correspondences_back_projection.restype = c_i64
        L.orc_correspondences_back_projection.argtypes = [c_f32p, c_i64, C.c_int, c_f32p, C.c_int, c_f32p, c_i64,
                                                          C.c_int, c_f32p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_point_normal.restype = C.c_int
        L.orc_point_normal.argtypes = [c_f32p, C.c_int, c_i32p, c_i64, c_f32p, c_f32p]
        L.orc_normals.restype = C.c_int
        L.orc_normals.argtypes = [C.c_void_p, c_f32p, C.c_int, c_f32p, c_i64, C.c_int, C.c_int, C.c_double, c_f32p,
                                  c_f32p, C.c_int]
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] in (3, 4), a.shape
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


def knn_brute(pts, queries, k):
    pts, queries = _f32(pts), _f32(queries)
    n, nq = len(pts), len(queries)
    ke = min(k, n)
    idx = np.empty((nq, max(ke, 0)), np.int32)
    d2 = np.empty((nq, max(ke, 0)), np.float32)
    if ke > 0 and nq > 0:
        lib().orc_knn_brute(_p(pts, c_f32p), n, pts.shape[1], _p(queries, c_f32p), nq, queries.shape[1], k,
                            _p(idx, c_i32p), _p(d2, c_f32p))
    return idx, d2


def _collect(offsets, pi, pd):
    total = int(offsets[-1])
    idx = np.ctypeslib.as_array(pi, shape=(max(total, 1),))[:total].copy()
    d2 = np.ctypeslib.as_array(pd, shape=(max(total, 1),))[:total].copy()
    lib().orc_free(pi)
    lib().orc_free(pd)
    return offsets, idx, d2


def radius_brute(pts, queries, radius, max_nn=0, sorted_results=True):
    """radius is the (double) radius; r2 = float(radius*radius) as impl/kdtree_flann.hpp:140."""
    pts, queries = _f32(pts), _f32(queries)
    nq = len(queries)
    offsets = np.zeros(nq + 1, np.int64)
    pi, pd = c_i32p(), c_f32p()
    r2 = np.float32(float(radius) * float(radius))
    lib().orc_radius_brute(_p(pts, c_f32p), len(pts), pts.shape[1], _p(queries, c_f32p), nq, queries.shape[1],
                           r2, max_nn, int(sorted_results), _p(offsets, c_i64p), C.byref(pi), C.byref(pd))
    return _collect(offsets, pi, pd)


class Tree:
    """FLANN-style single kd-tree (CPU baseline)."""

    def __init__(self, pts, leaf_max=15):
        self.pts = _f32(pts)
        self.h = lib().orc_tree_build(_p(self.pts, c_f32p), len(self.pts), self.pts.shape[1], leaf_max)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tree_free(self.h)
            self.h = None

    def knn(self, queries, k, nthreads=1):
        queries = _f32(queries)
        nq = len(queries)
        ke = min(k, len(self.pts))
        idx = np.empty((nq, ke), np.int32)
        d2 = np.empty((nq, ke), np.float32)
        if ke > 0 and nq > 0:
            lib().orc_tree_knn(self.h, _p(queries, c_f32p), nq, queries.shape[1], k, _p(idx, c_i32p),
                               _p(d2, c_f32p), nthreads)
        return idx, d2

    def radius(self, queries, radius, max_nn=0, sorted_results=True, nthreads=1):
        queries = _f32(queries)
        nq = len(queries)
        offsets = np.zeros(nq + 1, np.int64)
        pi, pd = c_i32p(), c_f32p()
        r2 = np.float32(float(radius) * float(radius))
        lib().orc_tree_radius(self.h, _p(queries, c_f32p), nq, queries.shape[1], r2, max_nn, int(sorted_results),
                              _p(offsets, c_i64p), C.byref(pi), C.byref(pd), nthreads)
        return _collect(offsets, pi, pd)


def estimate_point_to_plane_lls(src, tgt, tgt_normals, sidx=None, tidx=None, src_normals=None):
    """TransformationEstimationPointToPlaneLLS on paired points (index lists optional)."""
    src, tgt, tn = _f32(src), _f32(tgt), _f32(tgt_normals)
    sn = None if src_normals is None else _f32(src_normals)
    si = None if sidx is None else np.ascontiguousarray(sidx, np.int32)
    ti = None if tidx is None else np.ascontiguousarray(tidx, np.int32)
    n = len(si) if si is not None else (len(ti) if ti is not None else len(src))
    T = np.zeros(16, np.float32)
    lib().orc_estimate_point_to_plane_lls(_p(src, c_f32p), src.shape[1], None if sn is None else _p(sn, c_f32p),
                                          0 if sn is None else sn.shape[1], None if si is None else _p(si, c_i32p),
                                          _p(tgt, c_f32p), tgt.shape[1], _p(tn, c_f32p), tn.shape[1],
                                          None if ti is None else _p(ti, c_i32p), n, _p(T, c_f32p))
    return T.reshape(4, 4)


def correspondences_normal_shooting(src, src_normals, tgt, k=10, max_distance=float(np.finfo(np.float64).max)):
    src, sn, tgt = _f32(src), _f32(src_normals), _f32(tgt)
    out = np.zeros(len(src), CORR_DTYPE)
    n = lib().orc_correspondences_normal_shooting(_p(src, c_f32p), len(src), src.shape[1], _p(sn, c_f32p), sn.shape[1],
                                                  _p(tgt, c_f32p), len(tgt), tgt.shape[1], k, max_distance,
                                                  out.ctypes.data)
    return out[:n]


def correspondences_back_projection(src, src_normals, tgt, tgt_normals, k=10,
                                    max_distance=float(np.finfo(np.float64).max)):
    src, sn, tgt, tn = _f32(src), _f32(src_normals), _f32(tgt), _f32(tgt_normals)
    out = np.zeros(len(src), CORR_DTYPE)
    n = lib().orc_correspondences_back_projection(_p(src, c_f32p), len(src), src.shape[1], _p(sn, c_f32p), sn.shape[1],
                                                  _p(tgt, c_f32p), len(tgt), tgt.shape[1], _p(tn, c_f32p), tn.shape[1],
                                                  k, max_distance, out.ctypes.data)
    return out[:n]


def point_normal(pts, idx=None):
    """computePointNormal on an index list (None = the whole cloud): (plane[4], curvature, count)."""
    pts = _f32(pts)
    plane = np.zeros(4, np.float32)
    curv = np.zeros(1, np.float32)
    if idx is None:
        n, ip = len(pts), None
    else:
        ii = np.ascontiguousarray(idx, np.int32)
        n, ip = len(ii), _p(ii, c_i32p)
    cnt = lib().orc_point_normal(_p(pts, c_f32p), pts.shape[1], ip, n, _p(plane, c_f32p), _p(curv, c_f32p))
    return plane, float(curv[0]), cnt


def normals(tree, queries, k=0, radius=0.0, viewpoint=(0.0, 0.0, 0.0), nthreads=1):
    """NormalEstimation::computeFeature over `queries` against the cloud of `tree` (a Tree)."""
    queries = _f32(queries)
    out = np.full((len(queries), 4), np.nan, np.float32)
    vp = np.ascontiguousarray(viewpoint, np.float32)
    if len(queries):
        lib().orc_normals(tree.h, _p(tree.pts, c_f32p), tree.pts.shape[1], _p(queries, c_f32p), len(queries),
                          queries.shape[1], int(k), float(radius), _p(vp, c_f32p), _p(out, c_f32p), nthreads)
    return out


def transform(pts, T):
    out = _f32(pts).copy()
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    lib().orc_transform(_p(out, c_f32p), len(out), out.shape[1], _p(T, c_f32p))
    return out


def estimate_rigid_svd(src, tgt, sidx=None, tidx=None, scalar_double=False):
    src, tgt = _f32(src), _f32(tgt)
    n = len(sidx) if sidx is not None else len(src)
    si = np.ascontiguousarray(sidx, np.int32) if sidx is not None else None
    ti = np.ascontiguousarray(tidx, np.int32) if tidx is not None else None
    T = np.zeros(16, np.float32)
    fn = lib().orc_estimate_rigid_svd_f64 if scalar_double else lib().orc_estimate_rigid_svd
    fn(_p(src, c_f32p), src.shape[1], _p(si, c_i32p) if si is not None else None,
                                 _p(tgt, c_f32p), tgt.shape[1], _p(ti, c_i32p) if ti is not None else None,
                                 n, _p(T, c_f32p))
    return T.reshape(4, 4)


def correspondences(src, tgt, max_dist=float(np.sqrt(np.finfo(np.float64).max)), reciprocal=False):
    src, tgt = _f32(src), _f32(tgt)
    out = np.zeros(len(src), CORR_DTYPE)
    n = lib().orc_correspondences(_p(src, c_f32p), len(src), src.shape[1], _p(tgt, c_f32p), len(tgt),
                                  tgt.shape[1], max_dist, int(reciprocal), out.ctypes.data)
    return out[:n].copy()


def reject_distance(corr, max_dist):
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(corr), CORR_DTYPE)
    n = lib().orc_reject_distance(corr.ctypes.data, len(corr), max_dist, out.ctypes.data)
    return out[:n].copy()


def reject_surface_normal(corr, src_normals, tgt_normals, threshold=1.0):
    """CorrespondenceRejectorSurfaceNormal (impl/correspondence_rejection_surface_normal.hpp:44-66)."""
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    sn, tn = _f32(src_normals), _f32(tgt_normals)
    out = np.zeros(len(corr), CORR_DTYPE)
    n = lib().orc_reject_surface_normal(corr.ctypes.data, len(corr), _p(sn, c_f32p), sn.shape[1], _p(tn, c_f32p),
                                        tn.shape[1], float(threshold), out.ctypes.data)
    return out[:n].copy()


def reject_median(corr, factor=1.0):
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(corr), CORR_DTYPE)
    med = C.c_double(0)
    n = lib().orc_reject_median(corr.ctypes.data, len(corr), factor, out.ctypes.data, C.byref(med))
    return out[:n].copy(), med.value


def reject_one_to_one(corr):
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(corr), CORR_DTYPE)
    n = lib().orc_reject_one_to_one(corr.ctypes.data, len(corr), out.ctypes.data)
    return out[:n].copy()


def reject_trimmed(corr, overlap=0.5, min_corr=0):
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(corr), CORR_DTYPE)
    n = lib().orc_reject_trimmed(corr.ctypes.data, len(corr), overlap, min_corr, out.ctypes.data)
    return out[:n].copy()


def reject_var_trimmed(corr, min_ratio=0.05, max_ratio=0.95, lam=0.95, by_distance=False):
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(corr), CORR_DTYPE)
    f, d = C.c_double(0), C.c_double(0)
    n = lib().orc_reject_var_trimmed(corr.ctypes.data, len(corr), min_ratio, max_ratio, lam, int(by_distance),
                                     out.ctypes.data, C.byref(f), C.byref(d))
    return out[:n].copy(), f.value, d.value


def ransac_registration(src, tgt, sidx, tidx, inlier_threshold, max_iterations=1000):
    src, tgt = _f32(src), _f32(tgt)
    si = np.ascontiguousarray(sidx, np.int32)
    ti = np.ascontiguousarray(tidx, np.int32)
    pos = np.zeros(len(si), np.int32)
    T = np.zeros(16, np.float32)
    trials = C.c_int(0)
    n = lib().orc_ransac_registration(_p(src, c_f32p), len(src), src.shape[1], _p(tgt, c_f32p), len(tgt),
                                      tgt.shape[1], _p(si, c_i32p), _p(ti, c_i32p), len(si), inlier_threshold,
                                      max_iterations, _p(pos, c_i32p), _p(T, c_f32p), C.byref(trials))
    return (pos[:n].copy() if n >= 0 else None), T.reshape(4, 4), trials.value


def icp_default_params():
    p = IcpParams()
    lib().orc_icp_default_params(C.byref(p))
    return p


def icp_align(src, tgt, params=None, guess=None, nthreads=1):
    src, tgt = _f32(src), _f32(tgt)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.zeros((len(src), 4), np.float32)
    st = IcpStats()
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    rc = lib().orc_icp_align(C.byref(p), _p(src, c_f32p), len(src), src.shape[1], _p(tgt, c_f32p), len(tgt),
                             tgt.shape[1], _p(g, c_f32p) if g is not None else None, _p(T, c_f32p),
                             _p(out, c_f32p), C.byref(st), nthreads)
    return rc, T.reshape(4, 4), out, st


def icp_align_scaled(scale, src, tgt, params=None, guess=None, nthreads=1):
    """ICP with a rescaling point representation (registration.h:362-370): searches in scale * (x, y, z)."""
    src, tgt = _f32(src), _f32(tgt)
    sc = np.ascontiguousarray(scale, np.float32).reshape(3)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.zeros((len(src), 4), np.float32)
    st = IcpStats()
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    rc = lib().orc_icp_align_scaled(C.byref(p), _p(sc, c_f32p), _p(src, c_f32p), len(src), src.shape[1],
                                    _p(tgt, c_f32p), len(tgt), tgt.shape[1], _p(g, c_f32p) if g is not None else None,
                                    _p(T, c_f32p), _p(out, c_f32p), C.byref(st), nthreads)
    return rc, T.reshape(4, 4), out, st


def icp_align_point_to_plane(src, tgt, tgt_normals, params=None, guess=None, src_normals=None, nthreads=1):
    """ICP with TransformationEstimationPointToPlaneLLS as the estimator (icp.hpp:171)."""
    src, tgt, tn = _f32(src), _f32(tgt), _f32(tgt_normals)
    sn = _f32(src_normals) if src_normals is not None else None
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.zeros((len(src), 4), np.float32)
    st = IcpStats()
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    rc = lib().orc_icp_align_point_to_plane(
        C.byref(p), _p(src, c_f32p), len(src), src.shape[1], _p(sn, c_f32p) if sn is not None else None,
        sn.shape[1] if sn is not None else 0, _p(tgt, c_f32p), len(tgt), tgt.shape[1], _p(tn, c_f32p), tn.shape[1],
        _p(g, c_f32p) if g is not None else None, _p(T, c_f32p), _p(out, c_f32p), C.byref(st), nthreads)
    return rc, T.reshape(4, 4), out, st


def fitness_score(src, tgt, T, max_range=float(np.finfo(np.float64).max)):
    src, tgt = _f32(src), _f32(tgt)
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    return lib().orc_fitness_score(_p(src, c_f32p), len(src), src.shape[1], _p(tgt, c_f32p), len(tgt),
                                   tgt.shape[1], _p(T, c_f32p), max_range)
