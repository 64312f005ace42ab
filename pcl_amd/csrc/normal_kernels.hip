// normal_kernels.hip - surface normals on gfx950 (MI355X): the largest caller of the batched searches
// (SURVEY 8f rank 1).  Replaces NormalEstimation::computeFeature
// (features/include/pcl/features/impl/normal_3d.hpp:46-98) and the per-point search binding of
// Feature::compute (features/include/pcl/features/impl/feature.hpp:143-159): for every query point the
// neighbours come from the batched kNN / radius kernels (ONE search launch for the whole cloud instead of a
// per-point loop), then one lane per query gathers its neighbours, accumulates the nine moments
// (computeMeanAndCovarianceMatrix, common/include/pcl/common/impl/centroid.hpp:553-617) in fp64 about the
// query point - the reference's uncentred fp32 sums lose the covariance of clouds far from the origin -,
// solves for the smallest eigenpair (solvePlaneParameters, feature.hpp:47-86), takes the curvature
// |lambda0 / trace| and flips the normal towards the viewpoint (normal_3d.h:162-181).
#include "eig3.hpp"
#include "pclmi_internal.h"

namespace pclmi {
namespace {

// idx: CSR (offsets != null) or fixed pitch k rows.  out: nq x (nx, ny, nz, curvature).
__global__ __launch_bounds__(256) void normals_kernel(const float4 *__restrict__ cloud,
                                                      const float *__restrict__ q, int q_stride_f, int nq,
                                                      const int *__restrict__ idx,
                                                      const unsigned long long *__restrict__ offsets, int k, float vpx,
                                                      float vpy, float vpz, float4 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const float nanf_ = __builtin_nanf("");
  const float *qp = q + (size_t)i * q_stride_f;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  float4 res = make_float4(nanf_, nanf_, nanf_, nanf_);
  if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
    const unsigned long long b = offsets ? offsets[i] : (unsigned long long)i * (unsigned long long)k;
    const unsigned long long e = offsets ? offsets[i + 1] : b + (unsigned long long)k;
    double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
    double n = 0.0;
    for (unsigned long long a = b; a < e; ++a) {
      const int j = idx[a];
      if (j < 0) continue;
      const float4 p = cloud[j];
      const double dx = (double)p.x - (double)qx, dy = (double)p.y - (double)qy, dz = (double)p.z - (double)qz;
      n += 1.0;
      s1[0] += dx; s1[1] += dy; s1[2] += dz;
      s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
      s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
    }
    if (n > 0.0) {
      const double o[3] = {qx, qy, qz};
      double plane[4], curv;
      eig::plane_from_moments(n, s1, s2, o, plane, curv);
      // flipNormalTowardsViewpoint: towards (vp - point)
      const double c = ((double)vpx - qx) * plane[0] + ((double)vpy - qy) * plane[1] + ((double)vpz - qz) * plane[2];
      const double sgn = c < 0.0 ? -1.0 : 1.0;
      res = make_float4((float)(sgn * plane[0]), (float)(sgn * plane[1]), (float)(sgn * plane[2]), (float)curv);
    }
  }
  out[i] = res;
}

}  // namespace

void launch_normals(const pclmi_tree *t, const float *d_q, size_t qsf, size_t nq, const int *d_idx,
                    const uint64_t *d_offsets, int k, const float vp[3], float *d_out, hipStream_t s) {
  if (!nq) return;
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (const float4 *)t->d_cloud,
                     d_q, (int)qsf, (int)nq, d_idx, (const unsigned long long *)d_offsets, k, vp[0], vp[1], vp[2],
                     (float4 *)d_out);
}

}  // namespace pclmi
