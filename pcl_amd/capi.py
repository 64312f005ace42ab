"""ctypes binding of libpclmi.so (include/pclmi.h).  Plumbing only: every function here
forwards to one C-ABI entry point; numpy arrays are host buffers, torch CUDA tensors are
device buffers (passed by data_ptr()).
"""
import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libpclmi.so"
_LIB = None

STATUS = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "OOM", 5: "UNSUPPORTED", 6: "NOT_ENOUGH", 7: "COMM"}
ICP_NSUMS = 24
ICP_NSUMS_PLANE = 32  # sums of a point-to-plane context (pclmi.h "Point-to-plane ICP")

CORR_DTYPE = np.dtype([("index_query", np.int32), ("index_match", np.int32), ("distance", np.float32)])

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)
c_f64p = C.POINTER(C.c_double)


class PclmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pclmi: %s (%s)" % (msg, STATUS.get(code, code)))
        self.code = code


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("ransac_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("euclidean_fitness_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("ransac_outlier_threshold", C.c_double),
        ("min_correspondences", C.c_int),
    ]


class IcpStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("last_correspondences", C.c_int),
        ("last_inliers", C.c_int),
        ("fitness_delta", C.c_double),
        ("transform_delta", C.c_double),
        ("last_transform", C.c_float * 16),
    ]


# name -> (restype, argtypes); also the list the export test checks against include/pclmi.h
SIGNATURES = {
    "pclmi_version": (C.c_char_p, []),
    "pclmi_last_error": (C.c_char_p, []),
    "pclmi_device_count": (C.c_int, []),
    "pclmi_set_device": (C.c_int, [C.c_int]),
    "pclmi_timing_enable": (C.c_int, [C.c_int]),
    "pclmi_timing_read": (C.c_int, [C.c_int, c_f64p, c_u64p, C.c_int]),
    "pclmi_timing_pending": (C.c_int, []),
    "pclmi_tree_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                    C.POINTER(C.c_void_p)]),
    "pclmi_tree_create_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.POINTER(C.c_void_p)]),
    "pclmi_tree_destroy": (None, [C.c_void_p]),
    "pclmi_tree_size": (C.c_size_t, [C.c_void_p]),
    "pclmi_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                            C.POINTER(C.c_int)]),
    "pclmi_knn_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "pclmi_knn_dev_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "pclmi_knn_dev_pipelined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "pclmi_knn_dev_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "pclmi_knn_kernel_name": (C.c_char_p, [C.c_void_p, C.c_size_t, C.c_int]),
    "pclmi_radius_dev_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pclmi_radius": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_uint32, C.c_int,
                               C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pclmi_radius_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_uint32, C.c_int,
                                   C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_u64p, C.c_void_p]),
    "pclmi_free": (None, [C.c_void_p]),
    "pclmi_free_dev": (None, [C.c_void_p]),
    "pclmi_correspondences": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_double, C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_correspondences_reciprocal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                                   C.c_double, C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_reject_distance": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_reject_median": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.POINTER(C.c_size_t), c_f64p]),
    "pclmi_reject_one_to_one": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_reject_trimmed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_uint, C.c_void_p,
                                       C.POINTER(C.c_size_t)]),
    "pclmi_reject_var_trimmed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int,
                                           C.c_void_p, C.POINTER(C.c_size_t), c_f64p, c_f64p]),
    "pclmi_correspondences_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_double, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]),
    "pclmi_reject_surface_normal": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                              C.c_size_t, C.c_size_t, C.c_double, C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_reject_surface_normal_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                                  C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_void_p,
                                                  C.POINTER(C.c_size_t), C.c_void_p]),
    "pclmi_reject_distance_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.POINTER(C.c_size_t),
                                            C.c_void_p]),
    "pclmi_reject_median_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.POINTER(C.c_size_t),
                                          c_f64p, C.c_void_p]),
    "pclmi_reject_one_to_one_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]),
    "pclmi_reject_trimmed_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_uint, C.c_void_p,
                                           C.POINTER(C.c_size_t), C.c_void_p]),
    "pclmi_reject_var_trimmed_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int,
                                               C.c_void_p, C.POINTER(C.c_size_t), c_f64p, c_f64p, C.c_void_p]),
    "pclmi_estimate_rigid_svd_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                               C.c_void_p, C.c_size_t, c_f32p, C.c_void_p]),
    "pclmi_reject_sample_consensus": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_void_p,
                                                C.POINTER(C.c_size_t), c_f32p]),
    "pclmi_estimate_rigid_svd": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_size_t, C.c_void_p, C.c_size_t, c_f32p]),
    "pclmi_transform": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, c_f32p]),
    "pclmi_icp_default_params": (None, [C.POINTER(IcpParams)]),
    "pclmi_icp_align": (C.c_int, [C.POINTER(IcpParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p,
                                  c_f32p, C.c_void_p, C.POINTER(IcpStats)]),
    "pclmi_fitness_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p, C.c_double, c_f64p]),
    "pclmi_icp_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p, C.POINTER(C.c_void_p)]),
    "pclmi_icp_step_dev": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "pclmi_icp_step": (C.c_int, [C.c_void_p, C.c_double, c_f64p]),
    "pclmi_icp_solve": (C.c_int, [c_f64p, c_f32p]),
    "pclmi_icp_set_transform": (C.c_int, [C.c_void_p, c_f32p]),
    "pclmi_icp_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pclmi_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_double, c_f32p,
                                C.c_void_p]),
    "pclmi_normals_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_double, c_f32p,
                                    C.c_void_p, C.c_void_p]),
    "pclmi_point_normal": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, c_f32p, c_f32p]),
    "pclmi_estimate_point_to_plane_lls": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                    C.c_size_t, C.c_void_p, C.c_size_t, c_f32p]),
    "pclmi_correspondences_normal_shooting": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                        C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_double,
                                                        C.c_void_p, C.POINTER(C.c_size_t)]),
    "pclmi_correspondences_back_projection": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                        C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                        C.c_size_t, C.c_int, C.c_double, C.c_void_p,
                                                        C.POINTER(C.c_size_t)]),
    "pclmi_icp_set_criteria": (C.c_int, [C.c_void_p, C.POINTER(IcpParams)]),
    "pclmi_icp_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pclmi_icp_run_dev": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    "pclmi_icp_state": (C.c_int, [C.c_void_p, c_f32p, C.POINTER(IcpStats)]),
    "pclmi_icp_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pclmi_icp_begin_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p, C.c_int, C.c_int,
                                          C.POINTER(C.c_void_p)]),
    "pclmi_icp_shard_size": (C.c_size_t, [C.c_void_p]),
    "pclmi_icp_align_scaled": (C.c_int, [C.c_void_p, C.c_void_p, c_f32p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                         C.c_size_t, C.c_size_t, c_f32p, C.c_void_p, c_f32p, C.c_void_p, C.c_void_p]),
    "pclmi_icp_set_search_scale": (C.c_int, [C.c_void_p, c_f32p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "pclmi_icp_align_sharded": (C.c_int, [C.POINTER(IcpParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                          c_f32p, C.c_void_p, c_f32p, C.c_void_p, C.POINTER(IcpStats)]),
    "pclmi_icp_set_point_to_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_size_t]),
    "pclmi_icp_align_point_to_plane": (C.c_int, [C.POINTER(IcpParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                 C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p,
                                                 C.c_void_p, c_f32p, C.c_void_p, C.POINTER(IcpStats)]),
    "pclmi_icp_align_observed": (C.c_int, [C.POINTER(IcpParams), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, c_f32p,
                                           C.c_void_p, C.c_void_p, c_f32p, C.c_void_p, C.POINTER(IcpStats)]),
    "pclmi_comm_begin": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "pclmi_comm_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pclmi_comm_create_shm": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "pclmi_comm_rank": (C.c_int, [C.c_void_p]),
    "pclmi_comm_size": (C.c_int, [C.c_void_p]),
    "pclmi_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "pclmi_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pclmi_comm_status": (C.c_int, [C.c_void_p]),
    "pclmi_comm_destroy": (None, [C.c_void_p]),
}
COMM_HANDLE_BYTES = 128


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so with the same
    SONAME (libamdhip64.so.7) as the system ROCm one libpclmi.so is linked to; whichever is loaded first
    serves both.  torch does not find its GPUs when the system copy got in first, so when torch is
    installed its copy is loaded here (without importing torch); processes without torch use /opt/rocm."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        rt = Path(spec.origin).parent / "lib" / "libamdhip64.so"
        if rt.exists():
            try:
                C.CDLL(str(rt), mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load_library(path=None):
    """Loads libpclmi.so; raises if it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _share_hip_runtime_with_torch()
    import os
    p = Path(path) if path else Path(os.environ.get("PCLMI_LIB", LIB_PATH))  # PCLMI_LIB: development override
    if not p.exists():
        raise PclmiError(2, "%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C pcl_amd/csrc`" % p)
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PclmiError(rc, load_library().pclmi_last_error().decode())


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("points must be (n, 3) or (n, 4) float32, got %r" % (a.shape,))
    return a


def device_count():
    return load_library().pclmi_device_count()


TIMER_KNN, TIMER_RADIUS, TIMER_ICP_STEP, TIMER_ORDER = 0, 1, 2, 3


def timing_enable(on=True):
    _check(load_library().pclmi_timing_enable(int(on)))


def timing_read(which, reset=True):
    """Returns (total_ms, launches) of the HIP-event timer `which`."""
    ms, n = C.c_double(0), C.c_uint64(0)
    _check(load_library().pclmi_timing_read(which, C.byref(ms), C.byref(n), int(reset)))
    return ms.value, n.value


class Tree:
    """pcl::KdTreeFLANN::setInputCloud -> pclmi_tree_create."""

    def __init__(self, cloud, indices=None):
        lib = load_library()
        self.cloud = _pts(cloud)
        idx = None if indices is None else np.ascontiguousarray(indices, np.int32)
        h = C.c_void_p()
        _check(lib.pclmi_tree_create(self.cloud.ctypes.data, len(self.cloud), self.cloud.shape[1] * 4,
                                     None if idx is None else idx.ctypes.data, 0 if idx is None else len(idx), 0,
                                     C.byref(h)))
        self.h = h
        self._shards = 0  # live IcpShard contexts holding a raw pointer to this tree

    @classmethod
    def from_device(cls, xyz_ptr, n, stride_bytes, indices_ptr=None, n_indices=0, stream=None):
        """pclmi_tree_create_dev: the cloud (and the optional index list) already live in device memory."""
        self = cls.__new__(cls)
        self.cloud = None
        self._shards = 0
        h = C.c_void_p()
        _check(load_library().pclmi_tree_create_dev(xyz_ptr, n, stride_bytes, indices_ptr, n_indices, stream, C.byref(h)))
        self.h = h
        return self

    def debug_arrays(self):
        """(box[2M, 8], leaf[M, 64]) of the device tree (development aid, pclmi_debug_tree_download)."""
        lib = load_library()
        fn = lib.pclmi_debug_tree_download
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        m = C.c_int(0)
        _check(fn(self.h, C.byref(m), None, None))
        box = np.zeros((2 * m.value, 8), np.float32)
        leaf = np.zeros((m.value, 64), np.float32)
        if m.value:
            _check(fn(self.h, C.byref(m), box.ctypes.data, leaf.ctypes.data))
        return box, leaf

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_shards", 0) > 0:
                raise PclmiError(1, "Tree.close() while %d ICP context(s) still use the tree" % self._shards)
            load_library().pclmi_tree_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return load_library().pclmi_tree_size(self.h)

    def knn(self, queries, k):
        """Host buffers in / out.  Returns (idx[nq, k_eff], d2[nq, k_eff])."""
        q = _pts(queries)
        ke = min(k, len(self))
        idx = np.empty((len(q), ke), np.int32)
        d2 = np.empty((len(q), ke), np.float32)
        keff = C.c_int(0)
        _check(load_library().pclmi_knn(self.h, q.ctypes.data, len(q), q.shape[1] * 4, k, idx.ctypes.data,
                                        d2.ctypes.data, C.byref(keff)))
        assert keff.value == ke
        return idx, d2

    def knn_dev(self, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream=None):
        """Device pointers (ints) in / out; row pitch of the outputs is min(k, len(tree))."""
        _check(load_library().pclmi_knn_dev(self.h, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream))

    def knn_kernel_name(self, k, nq):
        return load_library().pclmi_knn_kernel_name(self.h, nq, k).decode()

    def knn_dev_async(self, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream=None):
        """As knn_dev, but only enqueues on `stream` (pclmi_knn_dev_async)."""
        _check(load_library().pclmi_knn_dev_async(self.h, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream))

    def knn_dev_pipelined(self, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream=None):
        """pclmi_knn_dev_pipelined: queries already valid at call time; the ordering pass overlaps the previous call."""
        _check(load_library().pclmi_knn_dev_pipelined(self.h, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream))

    def knn_dev_noalloc(self, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream=None):
        """pclmi_knn_dev_noalloc: enqueue-only, scratch in the library's static device array (no hipMalloc)."""
        _check(load_library().pclmi_knn_dev_noalloc(self.h, q_ptr, nq, q_stride_bytes, k, idx_ptr, d2_ptr, stream))

    def radius_dev_async(self, q_ptr, nq, q_stride_bytes, radius, offsets_ptr, idx_ptr, d2_ptr, capacity, status_ptr,
                         sorted_results=True, stream=None):
        """pclmi_radius_dev_async: enqueue-only, caller-owned result arrays of `capacity` entries."""
        _check(load_library().pclmi_radius_dev_async(self.h, q_ptr, nq, q_stride_bytes, float(radius),
                                                     int(sorted_results), offsets_ptr, idx_ptr, d2_ptr, capacity,
                                                     status_ptr, stream))

    def radius(self, queries, radius, max_nn=0, sorted_results=True):
        """Returns CSR (offsets[nq+1], idx, d2) as numpy arrays."""
        lib = load_library()
        q = _pts(queries)
        offsets = np.zeros(len(q) + 1, np.uint64)
        pi, pd = C.c_void_p(), C.c_void_p()
        _check(lib.pclmi_radius(self.h, q.ctypes.data, len(q), q.shape[1] * 4, float(radius), max_nn,
                                int(sorted_results), offsets.ctypes.data, C.byref(pi), C.byref(pd)))
        total = int(offsets[-1])
        if total:
            idx = np.ctypeslib.as_array(C.cast(pi, c_i32p), shape=(total,)).copy()
            d2 = np.ctypeslib.as_array(C.cast(pd, c_f32p), shape=(total,)).copy()
        else:
            idx, d2 = np.empty(0, np.int32), np.empty(0, np.float32)
        lib.pclmi_free(pi)
        lib.pclmi_free(pd)
        return offsets.astype(np.int64), idx, d2


def estimate_point_to_plane_lls(src, tgt, tgt_normals, src_idx=None, tgt_idx=None, src_normals=None):
    """TransformationEstimationPointToPlaneLLS: (status, T[4,4]); status 6 = singular system."""
    s, t, tn = _pts(src), _pts(tgt), _pts(tgt_normals)
    sn = None if src_normals is None else _pts(src_normals)
    si = None if src_idx is None else np.ascontiguousarray(src_idx, np.int32)
    ti = None if tgt_idx is None else np.ascontiguousarray(tgt_idx, np.int32)
    n = len(si) if si is not None else (len(ti) if ti is not None else len(s))
    T = np.zeros(16, np.float32)
    rc = load_library().pclmi_estimate_point_to_plane_lls(
        s.ctypes.data, len(s), s.shape[1] * 4, None if sn is None else sn.ctypes.data,
        0 if sn is None else sn.shape[1] * 4, None if si is None else si.ctypes.data, t.ctypes.data, len(t),
        t.shape[1] * 4, tn.ctypes.data, tn.shape[1] * 4, None if ti is None else ti.ctypes.data, n,
        T.ctypes.data_as(c_f32p))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4)


def correspondences_normal_shooting(target_tree, src, src_normals, k=10,
                                    max_distance=float(np.finfo(np.float64).max), src_indices=None):
    s, sn = _pts(src), _pts(src_normals)
    ii = None if src_indices is None else np.ascontiguousarray(src_indices, np.int32)
    out = np.zeros(len(s) if ii is None else len(ii), CORR_DTYPE)
    n = C.c_size_t(0)
    _check(load_library().pclmi_correspondences_normal_shooting(
        target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4, sn.ctypes.data, sn.shape[1] * 4,
        None if ii is None else ii.ctypes.data, 0 if ii is None else len(ii), int(k), float(max_distance),
        out.ctypes.data, C.byref(n)))
    return out[:n.value]


def correspondences_back_projection(target_tree, src, src_normals, tgt_normals, k=10,
                                    max_distance=float(np.finfo(np.float64).max), src_indices=None):
    s, sn, tn = _pts(src), _pts(src_normals), _pts(tgt_normals)
    ii = None if src_indices is None else np.ascontiguousarray(src_indices, np.int32)
    out = np.zeros(len(s) if ii is None else len(ii), CORR_DTYPE)
    n = C.c_size_t(0)
    _check(load_library().pclmi_correspondences_back_projection(
        target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4, sn.ctypes.data, sn.shape[1] * 4,
        tn.ctypes.data, len(tn), tn.shape[1] * 4,
        None if ii is None else ii.ctypes.data, 0 if ii is None else len(ii), int(k), float(max_distance),
        out.ctypes.data, C.byref(n)))
    return out[:n.value]


def normals(surface_tree, queries, k=0, radius=0.0, viewpoint=(0.0, 0.0, 0.0)):
    """pcl::NormalEstimation over `queries` against the surface tree: [nq, 4] = (nx, ny, nz, curvature)."""
    q = _pts(queries)
    out = np.full((len(q), 4), np.nan, np.float32)
    vp = np.ascontiguousarray(viewpoint, np.float32)
    _check(load_library().pclmi_normals(surface_tree.h, q.ctypes.data, len(q), q.shape[1] * 4, int(k), float(radius),
                                        vp.ctypes.data_as(c_f32p), out.ctypes.data))
    return out


def point_normal(points, indices=None):
    """pcl::computePointNormal: (plane[4], curvature) of one index list (None = the whole cloud)."""
    p = _pts(points)
    plane = np.zeros(4, np.float32)
    curv = C.c_float(0)
    ii = None if indices is None else np.ascontiguousarray(indices, np.int32)
    _check(load_library().pclmi_point_normal(p.ctypes.data, len(p), p.shape[1] * 4,
                                             None if ii is None else ii.ctypes.data, 0 if ii is None else len(ii),
                                             plane.ctypes.data_as(c_f32p), C.byref(curv)))
    return plane, curv.value


def correspondences(target_tree, src, max_distance=float(np.sqrt(np.finfo(np.float64).max)), src_indices=None):
    lib = load_library()
    s = _pts(src)
    si = None if src_indices is None else np.ascontiguousarray(src_indices, np.int32)
    cap = len(s) if si is None else len(si)
    out = np.zeros(cap, CORR_DTYPE)
    n = C.c_size_t(0)
    _check(lib.pclmi_correspondences(target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4,
                                     None if si is None else si.ctypes.data, 0 if si is None else len(si),
                                     max_distance, out.ctypes.data, C.byref(n)))
    return out[: n.value].copy()


def correspondences_reciprocal(target_tree, source_tree, src, tgt,
                               max_distance=float(np.sqrt(np.finfo(np.float64).max)), src_indices=None):
    lib = load_library()
    s, t = _pts(src), _pts(tgt)
    si = None if src_indices is None else np.ascontiguousarray(src_indices, np.int32)
    cap = len(s) if si is None else len(si)
    out = np.zeros(cap, CORR_DTYPE)
    n = C.c_size_t(0)
    _check(lib.pclmi_correspondences_reciprocal(target_tree.h, source_tree.h, s.ctypes.data, len(s), s.shape[1] * 4,
                                                None if si is None else si.ctypes.data,
                                                0 if si is None else len(si), t.ctypes.data, len(t), t.shape[1] * 4,
                                                max_distance, out.ctypes.data, C.byref(n)))
    return out[: n.value].copy()


def _rej(fn, corr, *args, extra=()):
    c = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(c), CORR_DTYPE)
    n = C.c_size_t(0)
    _check(fn(c.ctypes.data, len(c), *args, out.ctypes.data, C.byref(n), *extra))
    return out[: n.value].copy()


def reject_distance(corr, max_distance):
    return _rej(load_library().pclmi_reject_distance, corr, max_distance)


def reject_surface_normal(corr, src_normals, tgt_normals, threshold=1.0):
    """pcl::registration::CorrespondenceRejectorSurfaceNormal: pairs whose normals' dot product is > threshold."""
    sn, tn = _pts(src_normals), _pts(tgt_normals)
    return _rej(load_library().pclmi_reject_surface_normal, corr, sn.ctypes.data, len(sn), sn.shape[1] * 4,
                tn.ctypes.data, len(tn), tn.shape[1] * 4, float(threshold))


def reject_median(corr, factor=1.0):
    med = C.c_double(0)
    out = _rej(load_library().pclmi_reject_median, corr, factor, extra=(C.byref(med),))
    return out, med.value


def reject_one_to_one(corr):
    return _rej(load_library().pclmi_reject_one_to_one, corr)


def reject_trimmed(corr, overlap=0.5, min_corr=0):
    return _rej(load_library().pclmi_reject_trimmed, corr, overlap, min_corr)


def reject_var_trimmed(corr, min_ratio=0.05, max_ratio=0.95, lam=0.95, by_distance=False):
    """by_distance=False: the reference as shipped (the first #(sorted < trimmed) inputs); True: by own distance."""
    f, d = C.c_double(0), C.c_double(0)
    out = _rej(load_library().pclmi_reject_var_trimmed, corr, min_ratio, max_ratio, lam, int(by_distance),
               extra=(C.byref(f), C.byref(d)))
    return out, f.value, d.value


def reject_sample_consensus(src, tgt, corr, inlier_threshold=0.05, max_iterations=2**31 - 1):
    s, t = _pts(src), _pts(tgt)
    c = np.ascontiguousarray(corr, CORR_DTYPE)
    out = np.zeros(len(c), CORR_DTYPE)
    n = C.c_size_t(0)
    T = np.zeros(16, np.float32)
    _check(load_library().pclmi_reject_sample_consensus(s.ctypes.data, len(s), s.shape[1] * 4, t.ctypes.data, len(t),
                                                        t.shape[1] * 4, c.ctypes.data, len(c), inlier_threshold,
                                                        max_iterations, out.ctypes.data, C.byref(n),
                                                        T.ctypes.data_as(c_f32p)))
    return out[: n.value].copy(), T.reshape(4, 4)


def estimate_rigid_svd(src, tgt, src_idx=None, tgt_idx=None):
    lib = load_library()
    s, t = _pts(src), _pts(tgt)
    si = None if src_idx is None else np.ascontiguousarray(src_idx, np.int32)
    ti = None if tgt_idx is None else np.ascontiguousarray(tgt_idx, np.int32)
    n = len(si) if si is not None else (len(ti) if ti is not None else len(s))
    T = np.zeros(16, np.float32)
    _check(lib.pclmi_estimate_rigid_svd(s.ctypes.data, len(s), s.shape[1] * 4, None if si is None else si.ctypes.data,
                                        t.ctypes.data, len(t), t.shape[1] * 4, None if ti is None else ti.ctypes.data,
                                        n, T.ctypes.data_as(c_f32p)))
    return T.reshape(4, 4)


def transform(points, T):
    out = _pts(points).copy()
    Tm = np.ascontiguousarray(T, np.float32).reshape(16)
    _check(load_library().pclmi_transform(out.ctypes.data, len(out), out.shape[1] * 4, Tm.ctypes.data_as(c_f32p)))
    return out


def icp_default_params():
    p = IcpParams()
    load_library().pclmi_icp_default_params(C.byref(p))
    return p


def icp_align(target_tree, src, params=None, guess=None):
    """pcl::IterativeClosestPoint::align.  Returns (status, T[4,4], aligned[n,4], stats)."""
    lib = load_library()
    s = _pts(src)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.zeros((len(s), 4), np.float32)
    st = IcpStats()
    g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
    rc = lib.pclmi_icp_align(C.byref(p), target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4,
                             None if g is None else g.ctypes.data_as(c_f32p), T.ctypes.data_as(c_f32p),
                             out.ctypes.data, C.byref(st))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4), out, st


def icp_align_scaled(scaled_target_tree, scale, tgt, src, params=None, guess=None, comm=None):
    """pclmi_icp_align_scaled: ICP with a rescaling point representation (the tree indexes scale * target)."""
    lib = load_library()
    s, t = _pts(src), _pts(tgt)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.full((len(s), 4), np.nan, np.float32)
    st = IcpStats()
    sc = np.ascontiguousarray(scale, np.float32).reshape(3)
    g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
    rc = lib.pclmi_icp_align_scaled(C.byref(p), scaled_target_tree.h, sc.ctypes.data_as(c_f32p), t.ctypes.data, len(t),
                                    t.shape[1] * 4, s.ctypes.data, len(s), s.shape[1] * 4,
                                    None if g is None else g.ctypes.data_as(c_f32p), comm.h if comm else None,
                                    T.ctypes.data_as(c_f32p), out.ctypes.data, C.byref(st))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4), out, st


def icp_align_sharded(target_tree, src, comm, params=None, guess=None):
    """pclmi_icp_align_sharded: every rank passes the whole source.  Returns (status, T, aligned, stats); `aligned`
    holds this rank's own rows (the other rows are NaN)."""
    lib = load_library()
    s = _pts(src)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.full((len(s), 4), np.nan, np.float32)
    st = IcpStats()
    g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
    rc = lib.pclmi_icp_align_sharded(C.byref(p), target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4,
                                     None if g is None else g.ctypes.data_as(c_f32p), comm.h, T.ctypes.data_as(c_f32p),
                                     out.ctypes.data, C.byref(st))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4), out, st


def icp_align_point_to_plane(target_tree, src, tgt_normals, params=None, guess=None, src_normals=None, comm=None):
    """pclmi_icp_align_point_to_plane: ICP with TransformationEstimationPointToPlaneLLS.  tgt_normals: one row per row
    of the target cloud.  With a Comm of several ranks every rank passes the whole source and `aligned` holds the
    rank's own rows (the others NaN).  Returns (status, T[4,4], aligned[n,4], stats)."""
    lib = load_library()
    s = _pts(src)
    tn = _pts(tgt_normals)
    sn = None if src_normals is None else _pts(src_normals)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    sharded = comm is not None and comm.size > 1
    out = np.full((len(s), 4), np.nan if sharded else 0.0, np.float32)
    st = IcpStats()
    g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
    rc = lib.pclmi_icp_align_point_to_plane(
        C.byref(p), target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4, None if sn is None else sn.ctypes.data,
        0 if sn is None else sn.shape[1] * 4, tn.ctypes.data, len(tn), tn.shape[1] * 4,
        None if g is None else g.ctypes.data_as(c_f32p), None if comm is None else comm.h, T.ctypes.data_as(c_f32p),
        out.ctypes.data, C.byref(st))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4), out, st


ITERATION_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                           C.c_size_t)


def icp_align_observed(target_tree, src, on_iteration, params=None, guess=None, tgt_normals=None, src_normals=None):
    """pclmi_icp_align_observed: on_iteration(iteration, cloud[n, 4], src_idx[m], tgt_idx[m]) after every iteration
    (Registration::registerVisualizationCallback).  tgt_normals selects the point-to-plane estimator."""
    lib = load_library()
    s = _pts(src)
    tn = None if tgt_normals is None else _pts(tgt_normals)
    sn = None if src_normals is None else _pts(src_normals)
    p = params or icp_default_params()
    T = np.zeros(16, np.float32)
    out = np.zeros((len(s), 4), np.float32)
    st = IcpStats()
    g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
    n = len(s)

    def tramp(_user, it, cloud, si, ti, m):
        on_iteration(it, np.ctypeslib.as_array(cloud, shape=(n, 4)).copy(),
                     np.ctypeslib.as_array(si, shape=(m,)).copy() if m else np.zeros(0, np.int32),
                     np.ctypeslib.as_array(ti, shape=(m,)).copy() if m else np.zeros(0, np.int32))

    cb = ITERATION_FN(tramp)
    rc = lib.pclmi_icp_align_observed(
        C.byref(p), target_tree.h, s.ctypes.data, n, s.shape[1] * 4, None if sn is None else sn.ctypes.data,
        0 if sn is None else sn.shape[1] * 4, None if tn is None else tn.ctypes.data, 0 if tn is None else len(tn),
        0 if tn is None else tn.shape[1] * 4, None if g is None else g.ctypes.data_as(c_f32p),
        C.cast(cb, C.c_void_p), None, T.ctypes.data_as(c_f32p), out.ctypes.data, C.byref(st))
    if rc not in (0, 6):
        _check(rc)
    return rc, T.reshape(4, 4), out, st


def fitness_score(target_tree, src, T, max_range=float(np.finfo(np.float64).max)):
    s = _pts(src)
    Tm = np.ascontiguousarray(T, np.float32).reshape(16)
    score = C.c_double(0)
    _check(load_library().pclmi_fitness_score(target_tree.h, s.ctypes.data, len(s), s.shape[1] * 4,
                                              Tm.ctypes.data_as(c_f32p), max_range, C.byref(score)))
    return score.value


def icp_solve(sums):
    """Host-only: 24 fp64 sums -> row-major 4x4 (pclmi_icp_solve).  Needs no GPU."""
    s = np.ascontiguousarray(sums, np.float64)
    assert s.shape == (ICP_NSUMS,)
    T = np.zeros(16, np.float32)
    _check(load_library().pclmi_icp_solve(s.ctypes.data_as(c_f64p), T.ctypes.data_as(c_f32p)))
    return T.reshape(4, 4)


class Comm:
    """pclmi_comm: the multi-GPU exchange below the C ABI (one process per GPU, peer mailboxes over HIP IPC).

    Comm.from_shm(rank, world, name) exchanges the mailbox handles through a POSIX shared-memory object;
    Comm.from_allgather(rank, world, allgather) takes any callable bytes -> list of every rank's bytes
    (torch.distributed.all_gather_object, MPI ...)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_shm(cls, rank, world, name):
        h = C.c_void_p()
        _check(load_library().pclmi_comm_create_shm(rank, world, name.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_allgather(cls, rank, world, allgather):
        lib = load_library()
        h = C.c_void_p()
        mine = (C.c_ubyte * COMM_HANDLE_BYTES)()
        _check(lib.pclmi_comm_begin(rank, world, C.byref(h), mine))
        every = allgather(bytes(mine))
        assert len(every) == world and all(len(b) == COMM_HANDLE_BYTES for b in every)
        if world > 1:
            blob = b"".join(every)
            _check(lib.pclmi_comm_connect(h, blob))
        return cls(h)

    @property
    def rank(self):
        return load_library().pclmi_comm_rank(self.h)

    @property
    def size(self):
        return load_library().pclmi_comm_size(self.h)

    def set_timeout(self, seconds):
        _check(load_library().pclmi_comm_set_timeout(self.h, float(seconds)))

    def allreduce_f64(self, buf_ptr, count, stream=None):
        _check(load_library().pclmi_comm_allreduce_f64(self.h, buf_ptr, count, stream))

    def status(self):
        """Synchronises the device; raises PclmiError(COMM) if an exchange timed out."""
        _check(load_library().pclmi_comm_status(self.h))

    def close(self):
        if getattr(self, "h", None):
            load_library().pclmi_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IcpShard:
    """Per-rank ICP context for the sharded driver (pclmi_icp_begin / step / set_transform / end)."""

    def __init__(self, target_tree, src, guess=None, rank=0, world=1):
        """world > 1: `src` is the WHOLE source; the context keeps rank's contiguous run of its Hilbert order
        (pclmi_icp_begin_sharded) and end() returns the whole-size array with only that run's rows written."""
        lib = load_library()
        self.src = _pts(src)
        g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
        h = C.c_void_p()
        _check(lib.pclmi_icp_begin_sharded(target_tree.h, self.src.ctypes.data, len(self.src), self.src.shape[1] * 4,
                                           None if g is None else g.ctypes.data_as(c_f32p), rank, world, C.byref(h)))
        self.h = h
        self.n = len(self.src)
        self.shard_size = lib.pclmi_icp_shard_size(h)
        self.tree = target_tree  # the C context holds a raw pointer to the tree: keep it alive as long as the shard
        target_tree._shards += 1
        self.comm = None

    def set_comm(self, comm):
        """Attach a Comm: run_dev then exchanges the sums inside the step kernel (pclmi_icp_set_comm)."""
        _check(load_library().pclmi_icp_set_comm(self.h, None if comm is None else comm.h))
        self.comm = comm

    def set_point_to_plane(self, tgt_normals, src_normals=None):
        """Switch the context to TransformationEstimationPointToPlaneLLS (before the first step); its sums are then
        ICP_NSUMS_PLANE doubles (pclmi_icp_set_point_to_plane)."""
        tn = _pts(tgt_normals)
        sn = None if src_normals is None else _pts(src_normals)
        _check(load_library().pclmi_icp_set_point_to_plane(
            self.h, tn.ctypes.data, len(tn), tn.shape[1] * 4, None if sn is None else sn.ctypes.data,
            0 if sn is None else len(sn), 0 if sn is None else sn.shape[1] * 4))

    def step(self, max_correspondence_distance):
        sums = np.zeros(ICP_NSUMS, np.float64)
        _check(load_library().pclmi_icp_step(self.h, max_correspondence_distance, sums.ctypes.data_as(c_f64p)))
        return sums

    def step_dev(self, max_correspondence_distance, sums_ptr, stream=None):
        _check(load_library().pclmi_icp_step_dev(self.h, max_correspondence_distance, sums_ptr, stream))

    def set_transform(self, T):
        Tm = np.ascontiguousarray(T, np.float32).reshape(16)
        _check(load_library().pclmi_icp_set_transform(self.h, Tm.ctypes.data_as(c_f32p)))

    # device-resident iteration (no host round trip): see include/pclmi.h
    def set_criteria(self, params):
        _check(load_library().pclmi_icp_set_criteria(self.h, C.byref(params)))

    def solve_dev(self, sums_ptr, stream=None):
        _check(load_library().pclmi_icp_solve_dev(self.h, sums_ptr, stream))

    def run_dev(self, max_correspondence_distance, iterations, stream=None):
        _check(load_library().pclmi_icp_run_dev(self.h, max_correspondence_distance, iterations, stream))

    def state(self):
        """Synchronises; returns (status, final_T[4,4], stats)."""
        T = np.zeros(16, np.float32)
        st = IcpStats()
        rc = load_library().pclmi_icp_state(self.h, T.ctypes.data_as(c_f32p), C.byref(st))
        if rc not in (0, 6, 7):
            _check(rc)
        return rc, T.reshape(4, 4), st

    def end(self, want_aligned=True):
        out = np.zeros((self.n, 4), np.float32) if want_aligned else None
        h, self.h = self.h, None
        self.tree._shards -= 1
        _check(load_library().pclmi_icp_end(h, None if out is None else out.ctypes.data))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            try:
                self.tree._shards -= 1
                load_library().pclmi_icp_end(self.h, None)
            except Exception:
                pass
            self.h = None
