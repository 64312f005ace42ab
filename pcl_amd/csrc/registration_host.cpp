// registration_host.cpp - host-side math of the registration path (no GPU needed).
//
// The math itself lives in rigid_solve.hpp (shared with the device-side solve kernel).
// pclmi_icp_solve turns the PCLMI_ICP_NSUMS reduced fp64 sums of one ICP iteration into the rigid
// transform, i.e. the O(1) tail of TransformationEstimationSVD
// (registration/include/pcl/registration/impl/transformation_estimation_svd.hpp:114-168):
// centroids, H = sum (s - cs)(t - ct)', SVD, R = V U' with the determinant fix, t = ct - R cs.
// The reference does this in fp32 with a two-pass demean; here the sums arrive as one-pass fp64
// moments about a common origin, H = S_st - S_s S_t' / n, and the 3x3 SVD runs in fp64 - at least as
// accurate as the reference, within its 1e-5 tolerance.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pclmi_internal.h"
#include "rigid_solve.hpp"

namespace pclmi {

void svd3_f64(const double A[9], double U[9], double S[3], double V[9]) {
  double work[21];
  rs::svd3(A, U, S, V, work);
}

int solve_rigid_from_sums(const double sums[PCLMI_ICP_NSUMS], float T[16]) {
  double work[42];
  if (!rs::solve_from_sums(sums, T, work)) {
    set_error("pclmi_icp_solve: no correspondences (n = %g)", sums[0]);
    return PCLMI_ERR_NOT_ENOUGH;
  }
  return PCLMI_OK;
}

void mat4_mul_f32(const float A[16], const float B[16], float C[16]) { rs::mat4_mul(A, B, C); }

// TransformationEstimationPointToPlaneLLS: the 6x6 solve and constructTransformationMatrix live in rigid_solve.hpp
// (shared with the device loop).  sums: the 21 upper-triangle entries of ATA row by row, then the 6 of ATb.
int solve_point_to_plane(const double sums[27], float T[16]) {
  double work[42];
  if (!rs::solve_point_to_plane_sums(sums, T, work)) {
    set_error("point-to-plane estimation: the 6x6 normal equations are singular (degenerate geometry)");
    return PCLMI_ERR_NOT_ENOUGH;
  }
  return PCLMI_OK;
}

}  // namespace pclmi

extern "C" int pclmi_icp_solve(const double sums[PCLMI_ICP_NSUMS], float T[16]) {
  if (!sums || !T) {
    pclmi::set_error("pclmi_icp_solve: NULL argument");
    return PCLMI_ERR_INVALID;
  }
  return pclmi::solve_rigid_from_sums(sums, T);
}

extern "C" void pclmi_icp_default_params(pclmi_icp_params *p) {
  if (!p) return;
  p->max_iterations = 10;     // registration.h:97
  p->ransac_iterations = 1000;  // icp.h:104
  p->transformation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -1.7976931348623157e308;
  p->max_correspondence_distance = std::sqrt(1.7976931348623157e308);
  p->ransac_outlier_threshold = 0.05;
  p->min_correspondences = 3;
}
