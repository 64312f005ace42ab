"""The modular registration pipeline with everything resident in HBM (the _dev entry points):
CorrespondenceEstimation -> CorrespondenceRejector* -> TransformationEstimationSVD, the chain of
test/registration/test_registration_api.cpp:62-347, against (a) the reference's golden vectors on bun0 -> bun4,
(b) the host-pointer entry points and (c) the oracle, record for record."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from oracle import pcl_oracle as orc
    from pcl_amd import capi
    assert capi.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return capi, orc, torch, capi.load_library(), torch.device("cuda", 0)


class DevChain:
    """Thin driver over the _dev ABI: device arrays of pcl::Correspondence records as torch uint8 tensors."""

    def __init__(self, env):
        self.capi, self.orc, self.torch, self.lib, self.dev = env
        self.stream = self.torch.cuda.Stream(device=self.dev)

    def buf(self, n):
        return self.torch.empty(max(n, 1) * 12, dtype=self.torch.uint8, device=self.dev)

    def to_host(self, t, n):
        return np.frombuffer(t[:n * 12].cpu().numpy().tobytes(), dtype=self.capi.CORR_DTYPE).copy()

    def call(self, fn, d_in, n, *args, extra=()):
        out = self.buf(n)
        cnt = C.c_size_t(0)
        rc = fn(d_in.data_ptr(), n, *args, out.data_ptr(), C.byref(cnt), *extra, self.stream.cuda_stream)
        assert rc == 0, self.lib.pclmi_last_error()
        return out, cnt.value

    def correspondences(self, tree, d_src, n_src, stride, max_dist, d_rows=None, n_rows=0):
        out = self.buf(n_rows if d_rows is not None else n_src)
        cnt = C.c_size_t(0)
        rc = self.lib.pclmi_correspondences_dev(tree.h, d_src.data_ptr(), n_src, stride,
                                                None if d_rows is None else d_rows.data_ptr(), n_rows, max_dist,
                                                out.data_ptr(), C.byref(cnt), self.stream.cuda_stream)
        assert rc == 0, self.lib.pclmi_last_error()
        return out, cnt.value


def test_bun_chain_in_hbm_equals_goldens_and_host_path(env):
    capi, orc, torch, lib, dev = env
    ch = DevChain(env)
    api = util.golden("registration_api.json")
    src, tgt = util.bun0(), util.bun4()
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    tree = capi.Tree(tgt)
    big = float(np.sqrt(np.finfo(np.float64).max))
    corr_d, n = ch.correspondences(tree, d_src, len(src), 12, big)
    corr = ch.to_host(corr_d, n)
    exp = np.array(api["correspondences_original"])
    assert n == 397 and corr["index_match"].tolist() == exp[:, 1].tolist()
    assert np.array_equal(corr, capi.correspondences(tree, src))
    # distance rejector -> 97 golden pairs
    rd_d, nrd = ch.call(lib.pclmi_reject_distance_dev, corr_d, n, C.c_float(api["rej_dist_max_dist"]))
    rd = ch.to_host(rd_d, nrd)
    exp = np.array(api["correspondences_dist"])
    assert rd["index_query"].tolist() == exp[:, 0].tolist() and rd["index_match"].tolist() == exp[:, 1].tolist()
    assert np.array_equal(rd, capi.reject_distance(corr, api["rej_dist_max_dist"]))
    # one-to-one -> 103 golden pairs; trimmed -> 198
    oo_d, noo = ch.call(lib.pclmi_reject_one_to_one_dev, corr_d, n)
    assert np.array_equal(ch.to_host(oo_d, noo), capi.reject_one_to_one(corr)) and noo == len(api["correspondences_one_to_one"])
    tr_d, ntr = ch.call(lib.pclmi_reject_trimmed_dev, corr_d, n, C.c_float(api["rej_trimmed_overlap"]), 0)
    assert np.array_equal(ch.to_host(tr_d, ntr), capi.reject_trimmed(corr, api["rej_trimmed_overlap"]))
    assert ntr == len(api["correspondences_trimmed"])
    # median, var-trimmed (both selections)
    med = C.c_double(0)
    md_d, nmd = ch.call(lib.pclmi_reject_median_dev, corr_d, n, C.c_double(0.5), extra=(C.byref(med),))
    hm, hmed = capi.reject_median(corr, 0.5)
    assert med.value == hmed and np.array_equal(ch.to_host(md_d, nmd), hm)
    for by in (0, 1):
        f, d = C.c_double(0), C.c_double(0)
        vt_d, nvt = ch.call(lib.pclmi_reject_var_trimmed_dev, corr_d, n, C.c_double(0.05), C.c_double(0.95),
                            C.c_double(0.95), by, extra=(C.byref(f), C.byref(d)))
        hv, hf, hd = capi.reject_var_trimmed(corr, by_distance=bool(by))
        assert (f.value, d.value) == (hf, hd) and np.array_equal(ch.to_host(vt_d, nvt), hv)
    # TransformationEstimationSVD on device-resident correspondences; the reference's transform_from_SVD golden is
    # taken on the 53 reciprocal pairs (test_registration_api.cpp:325-347)
    T = np.zeros(16, np.float32)
    rc = lib.pclmi_estimate_rigid_svd_dev(d_src.data_ptr(), len(src), 12, d_tgt.data_ptr(), len(tgt), 12,
                                          corr_d.data_ptr(), n, T.ctypes.data_as(capi.c_f32p), ch.stream.cuda_stream)
    assert rc == 0
    Th = capi.estimate_rigid_svd(src, tgt, corr["index_query"], corr["index_match"])
    assert np.abs(T.reshape(4, 4) - Th).max() <= 1e-6
    rec = capi.correspondences_reciprocal(tree, capi.Tree(src), src, tgt)
    assert len(rec) == len(api["correspondences_reciprocal"])
    rec_d = torch.from_numpy(np.frombuffer(rec.tobytes(), np.uint8).copy()).to(dev)
    rc = lib.pclmi_estimate_rigid_svd_dev(d_src.data_ptr(), len(src), 12, d_tgt.data_ptr(), len(tgt), 12,
                                          rec_d.data_ptr(), len(rec), T.ctypes.data_as(capi.c_f32p),
                                          ch.stream.cuda_stream)
    assert rc == 0 and np.allclose(T.reshape(4, 4), np.array(api["transform_from_SVD"]), atol=1e-4)


@pytest.mark.parametrize("n", [1, 70, 2049, 300_000])
def test_chain_on_synthetic_clouds_vs_oracle(env, n):
    capi, orc, torch, lib, dev = env
    ch = DevChain(env)
    rng = np.random.default_rng(n)
    tgt = (rng.random((max(n, 50), 3), dtype=np.float32) * 10).astype(np.float32)
    src = np.zeros((n, 4), np.float32)  # PointXYZ rows (16-byte stride), some non-finite
    src[:, :3] = tgt[rng.integers(0, len(tgt), n)] + rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    tree = capi.Tree(tgt)
    rows = np.sort(rng.choice(n, max(1, n // 2), replace=False)).astype(np.int32)
    d_rows = torch.from_numpy(rows).to(dev)
    if n > 10:  # non-finite source rows and an index view: device path == host path (itself tested against the oracle)
        bad = src.copy()
        bad[rng.integers(0, n, n // 50 + 1), 1] = np.nan
        d_bad = torch.from_numpy(bad).to(dev)
        for use_rows in (False, True):
            corr_d, cnt = ch.correspondences(tree, d_bad, n, 16, 0.08, d_rows if use_rows else None, len(rows))
            assert np.array_equal(ch.to_host(corr_d, cnt), capi.correspondences(tree, bad, 0.08, rows if use_rows else None))
    d_src = torch.from_numpy(src).to(dev)
    corr_d, cnt = ch.correspondences(tree, d_src, n, 16, 0.08)
    got = ch.to_host(corr_d, cnt)
    assert np.array_equal(got, orc.correspondences(src[:, :3], tgt, 0.08))
    if cnt == 0:
        return
    out_d, c2 = ch.call(lib.pclmi_reject_distance_dev, corr_d, cnt, C.c_float(0.05))
    assert np.array_equal(ch.to_host(out_d, c2), orc.reject_distance(got, 0.05))
    med = C.c_double(0)
    out_d, c3 = ch.call(lib.pclmi_reject_median_dev, corr_d, cnt, C.c_double(1.5), extra=(C.byref(med),))
    o, om = orc.reject_median(got, 1.5)
    assert med.value == om and np.array_equal(ch.to_host(out_d, c3), o)
    out_d, c4 = ch.call(lib.pclmi_reject_one_to_one_dev, corr_d, cnt)
    assert np.array_equal(ch.to_host(out_d, c4), orc.reject_one_to_one(got))
    # chained in HBM: one-to-one -> trimmed -> SVD estimate, counts are the only thing the host sees
    tr_d, c5 = ch.call(lib.pclmi_reject_trimmed_dev, out_d, c4, C.c_float(0.7), 3)
    kept = ch.to_host(tr_d, c5)
    assert np.array_equal(kept, orc.reject_trimmed(orc.reject_one_to_one(got), 0.7, 3))
    d_tgt = torch.from_numpy(tgt).to(dev)
    T = np.zeros(16, np.float32)
    rc = lib.pclmi_estimate_rigid_svd_dev(d_src.data_ptr(), n, 16, d_tgt.data_ptr(), len(tgt), 12, tr_d.data_ptr(), c5,
                                          T.ctypes.data_as(capi.c_f32p), ch.stream.cuda_stream)
    if c5 >= 3:
        assert rc == 0
        To = orc.estimate_rigid_svd(src[:, :3], tgt, kept["index_query"], kept["index_match"], scalar_double=True)
        assert np.abs(T.reshape(4, 4) - To).max() <= 1e-5
