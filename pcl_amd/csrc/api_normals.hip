// api_normals.hip - C ABI: surface normals (NormalEstimation) on top of the batched searches.
#include <cmath>
#include <cstdlib>
#include <limits>
#include <vector>

#include "eig3.hpp"
#include "pclmi_internal.h"

using namespace pclmi;

extern "C" {

int pclmi_normals_dev(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k, double radius,
                      const float viewpoint[3], float *d_normals, void *stream) {
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_INVALID, "tree is NULL");
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  PCLMI_CHECK((k > 0) != (radius > 0.0), PCLMI_ERR_INVALID,
              "exactly one of k (%d) and radius (%g) must be positive (setKSearch / setRadiusSearch)", k, radius);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  if (nq == 0) return PCLMI_OK;
  PCLMI_CHECK(d_q && d_normals, PCLMI_ERR_INVALID, "NULL query/output pointer");
  PCLMI_CHECK(nq < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 queries in one call");
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  const float zero[3] = {0.f, 0.f, 0.f};
  const float *vp = viewpoint ? viewpoint : zero;
  const size_t qsf = q_stride_bytes / 4;
  if (t->n == 0) {  // no neighbours anywhere: every normal is NaN (normal_3d.hpp:62-68)
    launch_normals(t, d_q, qsf, nq, nullptr, nullptr, 0, vp, d_normals, s);
    PCLMI_HIP(hipStreamSynchronize(s));
    return PCLMI_OK;
  }
  if (k > 0) {
    const int k_eff = (size_t)k > t->n ? (int)t->n : k;
    static const bool unfused = [] {  // development switch: PCLMI_NORMALS=split runs search and plane fit separately
      const char *e = getenv("PCLMI_NORMALS");
      return e && e[0] == 's';
    }();
    if (knn_normals_fused(k_eff) && !unfused) {
      // ONE kernel: the lane that finds a point's k neighbours fits the plane (knn_kernels.hip normals_finish); the
      // neighbour lists never reach HBM
      WsGuard g(t, s);
      PCLMI_CHECK(g.w, PCLMI_ERR_OOM, "no workspace");
      const int32_t *order = nullptr;
      RC(launch_morton_order(t, g.w, d_q, nq, qsf, nullptr, nullptr, s, &order));
      RC(launch_knn_normals(t, g.w, d_q, nq, qsf, order, k_eff, vp, d_normals, s));
      PCLMI_HIP(hipStreamSynchronize(s));
      return PCLMI_OK;
    }
    // neighbour lists of the whole batch: one search launch (the reference loops over points, feature.hpp:188-220)
    // (buffers come from the library's result-buffer cache: no hipMalloc / hipFree in the steady state)
    int *idx = (int *)dev_cache_get(nq * (size_t)k_eff * sizeof(int));
    float *d2 = (float *)dev_cache_get(nq * (size_t)k_eff * sizeof(float));
    int rc = (idx && d2) ? pclmi_knn_dev(t, d_q, nq, q_stride_bytes, k_eff, idx, d2, s) : (int)PCLMI_ERR_OOM;
    hipError_t e = hipSuccess;
    if (rc == PCLMI_OK) {
      launch_normals(t, d_q, qsf, nq, idx, nullptr, k_eff, vp, d_normals, s);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    dev_cache_put(idx);
    dev_cache_put(d2);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "normals", __FILE__, __LINE__);
    return PCLMI_OK;
  }
  static const bool split_radius = [] {  // development switch, as above
    const char *e = getenv("PCLMI_NORMALS");
    return e && e[0] == 's';
  }();
  if (!split_radius) {
    // ONE traversal: hits are folded into the plane-fit moments as they are found (radius_kernels.hip); the
    // variable-length neighbour lists (count pass, scan, fill, sort, CSR arrays) are never built
    WsGuard g(t, s);
    PCLMI_CHECK(g.w, PCLMI_ERR_OOM, "no workspace");
    RC(radius_normals_dev(t, g.w, d_q, nq, qsf, (float)(radius * radius), vp, d_normals, s));  // kdtree_flann.hpp:140
    PCLMI_HIP(hipStreamSynchronize(s));
    return PCLMI_OK;
  }
  DevBuf<uint64_t> off;
  RC(off.alloc(nq + 1));
  int32_t *ri = nullptr;
  float *rd = nullptr;
  uint64_t total = 0;
  RC(pclmi_radius_dev(t, d_q, nq, q_stride_bytes, radius, 0, 1, off.p, &ri, &rd, &total, s));
  launch_normals(t, d_q, qsf, nq, ri, off.p, 0, vp, d_normals, s);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  pclmi_free_dev(ri);
  pclmi_free_dev(rd);
  if (e != hipSuccess) return hip_fail(e, "normals", __FILE__, __LINE__);
  return PCLMI_OK;
}

int pclmi_normals(const pclmi_tree *t, const float *q, size_t nq, size_t q_stride_bytes, int k, double radius,
                  const float viewpoint[3], float *normals) {
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_INVALID, "tree is NULL");
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  if (nq == 0) return PCLMI_OK;
  PCLMI_CHECK(q && normals, PCLMI_ERR_INVALID, "NULL query/output pointer");
  PCLMI_HIP(hipSetDevice(t->device));
  DevBuf<unsigned char> dq;
  DevBuf<float> dn;
  RC(dq.alloc(nq * q_stride_bytes));
  RC(dn.alloc(nq * 4));
  PCLMI_HIP(hipMemcpy(dq.p, q, nq * q_stride_bytes, hipMemcpyHostToDevice));
  RC(pclmi_normals_dev(t, (const float *)dq.p, nq, q_stride_bytes, k, radius, viewpoint, dn.p, nullptr));
  PCLMI_HIP(hipMemcpy(normals, dn.p, nq * 4 * sizeof(float), hipMemcpyDeviceToHost));
  return PCLMI_OK;
}

int pclmi_point_normal(const float *pts, size_t n_rows, size_t stride_bytes, const int32_t *indices, size_t n_indices,
                       float plane[4], float *curvature) {
  PCLMI_CHECK(plane && curvature, PCLMI_ERR_INVALID, "NULL output");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  const float nan = std::numeric_limits<float>::quiet_NaN();
  plane[0] = plane[1] = plane[2] = plane[3] = *curvature = nan;
  const size_t n = indices ? n_indices : n_rows;
  PCLMI_CHECK(n == 0 || pts, PCLMI_ERR_INVALID, "pts is NULL");
  const size_t sf = stride_bytes / 4;
  // a single list is O(n) host work (the batched device path is pclmi_normals); moments about the first finite point
  double o[3] = {0, 0, 0}, s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0}, cnt = 0.0;
  bool have_o = false;
  for (size_t a = 0; a < n; ++a) {
    const size_t row = indices ? (size_t)indices[a] : a;
    PCLMI_CHECK(row < n_rows, PCLMI_ERR_INVALID, "index %zu outside the cloud (%zu rows)", row, n_rows);
    const float *p = pts + row * sf;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    if (!have_o) {
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
      have_o = true;
    }
    const double dx = p[0] - o[0], dy = p[1] - o[1], dz = p[2] - o[2];
    cnt += 1.0;
    s1[0] += dx; s1[1] += dy; s1[2] += dz;
    s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
    s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
  }
  if (cnt == 0.0) return PCLMI_OK;  // computePointNormal: NaN plane for an empty neighbourhood (normal_3d.h:96-101)
  double pl[4], cv;
  eig::plane_from_moments(cnt, s1, s2, o, pl, cv);
  for (int d = 0; d < 4; ++d) plane[d] = (float)pl[d];
  *curvature = (float)cv;
  return PCLMI_OK;
}

}  // extern "C"
