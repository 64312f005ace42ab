// eig3.hpp - smallest eigenpair of a symmetric 3x3 in fp64 by cyclic two-sided Jacobi rotations, as
// host+device inline code with compile-time indices (registers only on the GPU).  Stands in for pcl::eigen33
// (common/include/pcl/common/eigen.h:334-368) behind solvePlaneParameters
// (features/include/pcl/features/impl/feature.hpp:47-86): the reference's closed-form fp32 solver is accurate
// to ~1e-5 relative; this one to fp64 rounding.
#ifndef PCLMI_EIG3_HPP_
#define PCLMI_EIG3_HPP_

#include <hip/hip_runtime.h>

namespace pclmi {
namespace eig {

// rotation annihilating a[P][Q]; R is the third index
template <int P, int Q, int R>
__host__ __device__ inline bool rotate(double (&a)[9], double (&v)[9]) {
  const double apq = a[P * 3 + Q];
  const double app = a[P * 3 + P], aqq = a[Q * 3 + Q];
  if (apq == 0.0 || fabs(apq) <= 1e-16 * (fabs(app) + fabs(aqq))) return false;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  a[P * 3 + P] = app - t * apq;
  a[Q * 3 + Q] = aqq + t * apq;
  a[P * 3 + Q] = a[Q * 3 + P] = 0.0;
  const double arp = a[R * 3 + P], arq = a[R * 3 + Q];
  a[R * 3 + P] = a[P * 3 + R] = c * arp - s * arq;
  a[R * 3 + Q] = a[Q * 3 + R] = s * arp + c * arq;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double vp = v[r * 3 + P], vq = v[r * 3 + Q];
    v[r * 3 + P] = c * vp - s * vq;
    v[r * 3 + Q] = s * vp + c * vq;
  }
  return true;
}

// cov: full symmetric 3x3 (row-major).  lambda0 = smallest eigenvalue, vec = its unit eigenvector.
__host__ __device__ inline void smallest(const double cov[9], double &lambda0, double vec[3]) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = cov[i];
  for (int sweep = 0; sweep < 16; ++sweep) {
    bool rot = rotate<0, 1, 2>(a, v);
    rot |= rotate<0, 2, 1>(a, v);
    rot |= rotate<1, 2, 0>(a, v);
    if (!rot) break;
  }
  const double e0 = a[0], e1 = a[4], e2 = a[8];
  const bool c1 = e1 < e0;               // ties keep the lower axis
  const double m01 = c1 ? e1 : e0;
  const bool c2 = e2 < m01;
  lambda0 = c2 ? e2 : m01;
#pragma unroll
  for (int r = 0; r < 3; ++r) vec[r] = c2 ? v[r * 3 + 2] : (c1 ? v[r * 3 + 1] : v[r * 3 + 0]);
  const double n = sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
  if (n > 0.0) {
    vec[0] /= n;
    vec[1] /= n;
    vec[2] /= n;
  }
}

// From the moments of n points about any origin o: s1 = sum (p - o), s2 = sum (p - o)(p - o)' (xx xy xz yy yz zz).
// plane = (nx, ny, nz, d) with d = -n . centroid; curvature = |lambda0 / trace| (feature.hpp:80-85).
__host__ __device__ inline void plane_from_moments(double n, const double s1[3], const double s2[6], const double o[3],
                                                   double plane[4], double &curvature) {
  const double mx = s1[0] / n, my = s1[1] / n, mz = s1[2] / n;
  double c[9];
  c[0] = s2[0] / n - mx * mx;
  c[1] = c[3] = s2[1] / n - mx * my;
  c[2] = c[6] = s2[2] / n - mx * mz;
  c[4] = s2[3] / n - my * my;
  c[5] = c[7] = s2[4] / n - my * mz;
  c[8] = s2[5] / n - mz * mz;
  double l0, v[3];
  smallest(c, l0, v);
  const double tr = c[0] + c[4] + c[8];
  curvature = tr != 0.0 ? fabs(l0 / tr) : 0.0;
  plane[0] = v[0];
  plane[1] = v[1];
  plane[2] = v[2];
  plane[3] = -(v[0] * (mx + o[0]) + v[1] * (my + o[1]) + v[2] * (mz + o[2]));
}

}  // namespace eig
}  // namespace pclmi
#endif
