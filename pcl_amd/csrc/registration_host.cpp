// registration_host.cpp - host-side math of the registration path (no GPU needed).
//
// pclmi_icp_solve turns the PCLMI_ICP_NSUMS reduced fp64 sums of one ICP iteration into the rigid
// transform, i.e. the O(1) tail of TransformationEstimationSVD
// (registration/include/pcl/registration/impl/transformation_estimation_svd.hpp:114-168):
// centroids, H = sum (s - cs)(t - ct)', SVD, R = V U' with the determinant fix, t = ct - R cs.
// The reference does this in fp32 with a two-pass demean; here the sums arrive as one-pass fp64
// moments about a common origin, H = S_st - S_s S_t' / n, and the 3x3 SVD runs in fp64 - at least as
// accurate as the reference, within its 1e-5 tolerance.
#include <cmath>
#include <cstring>

#include "pclmi_internal.h"

namespace pclmi {

// one-sided Jacobi SVD of a 3x3, singular values descending (as Eigen::JacobiSVD orders them, so
// that "column 2" of the determinant fix is the smallest singular direction).  A = U diag(S) V'.
void svd3_f64(const double A[9], double U[9], double S[3], double V[9]) {
  double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(a, A, sizeof(a));
  for (int sweep = 0; sweep < 100; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += a[r * 3 + p] * a[r * 3 + p];
          beta += a[r * 3 + q] * a[r * 3 + q];
          gamma += a[r * 3 + p] * a[r * 3 + q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          double x = a[r * 3 + p], y = a[r * 3 + q];
          a[r * 3 + p] = c * x - s * y;
          a[r * 3 + q] = s * x + c * y;
          x = v[r * 3 + p];
          y = v[r * 3 + q];
          v[r * 3 + p] = c * x - s * y;
          v[r * 3 + q] = s * x + c * y;
        }
      }
    if (!rotated) break;
  }
  double s[3];
  for (int c = 0; c < 3; ++c) s[c] = std::sqrt(a[c] * a[c] + a[3 + c] * a[3 + c] + a[6 + c] * a[6 + c]);
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (s[ord[j]] > s[ord[i]]) std::swap(ord[i], ord[j]);
  for (int c = 0; c < 3; ++c) {
    S[c] = s[ord[c]];
    for (int r = 0; r < 3; ++r) {
      V[r * 3 + c] = v[r * 3 + ord[c]];
      U[r * 3 + c] = S[c] > 0.0 ? a[r * 3 + ord[c]] / S[c] : 0.0;
    }
  }
  // rank-deficient H (three-point RANSAC samples are rank 2): complete U to an orthonormal basis,
  // which is what ComputeFullU returns.
  const double tiny = S[0] * 1e-13;
  auto col = [&](double *M, int c, double out[3]) { out[0] = M[c]; out[1] = M[3 + c]; out[2] = M[6 + c]; };
  auto setcol = [&](double *M, int c, const double in[3]) { M[c] = in[0]; M[3 + c] = in[1]; M[6 + c] = in[2]; };
  if (S[2] <= tiny) {
    double u0[3], u1[3];
    col(U, 0, u0);
    if (S[0] <= 0.0) {
      u0[0] = 1; u0[1] = 0; u0[2] = 0;
      setcol(U, 0, u0);
    }
    if (S[1] <= tiny) {
      int m = 0;
      for (int d = 1; d < 3; ++d)
        if (std::fabs(u0[d]) < std::fabs(u0[m])) m = d;
      double e[3] = {0, 0, 0};
      e[m] = 1.0;
      double w[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
      const double nw = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      for (int d = 0; d < 3; ++d) w[d] /= nw;
      setcol(U, 1, w);
    }
    col(U, 1, u1);
    const double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2],
                          u0[0] * u1[1] - u0[1] * u1[0]};
    setcol(U, 2, u2);
  }
}

static double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

int solve_rigid_from_sums(const double sums[PCLMI_ICP_NSUMS], float T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  const double n = sums[0];
  if (!(n >= 1.0)) {
    set_error("pclmi_icp_solve: no correspondences (n = %g)", n);
    return PCLMI_ERR_NOT_ENOUGH;
  }
  double o[3] = {0, 0, 0};
  if (sums[18] > 0.0)
    for (int d = 0; d < 3; ++d) o[d] = sums[19 + d] / sums[18];
  double ms[3], mt[3];  // centroids relative to the origin
  for (int d = 0; d < 3; ++d) {
    ms[d] = sums[1 + d] / n;
    mt[d] = sums[4 + d] / n;
  }
  double H[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) H[r * 3 + c] = sums[7 + r * 3 + c] - n * ms[r] * mt[c];
  double U[9], S[3], V[9];
  svd3_f64(H, U, S, V);
  if (det3(U) * det3(V) < 0)  // transformation_estimation_svd.hpp:156-160
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += V[r * 3 + k] * U[c * 3 + k];  // V * U'
      R[r * 3 + c] = s;
    }
  for (int r = 0; r < 3; ++r) {
    double rc = 0;
    for (int c = 0; c < 3; ++c) rc += R[r * 3 + c] * (ms[c] + o[c]);
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = (float)R[r * 3 + c];
    T[r * 4 + 3] = (float)((mt[r] + o[r]) - rc);  // t = ct - R cs
  }
  return PCLMI_OK;
}

void mat4_mul_f32(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      R[r * 4 + c] = s;
    }
  std::memcpy(C, R, sizeof(R));
}

// eigenvalues of a symmetric 3x3 by cyclic Jacobi (stands in for pcl::eigen33 in
// SampleConsensusModelRegistration::computeSampleDistanceThreshold, sac_model_registration.h:257-279)
void sym_eigvals3_f64(const double Min[9], double ev[3]) {
  double a[9];
  std::memcpy(a, Min, sizeof(a));
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = std::fabs(a[1]) + std::fabs(a[2]) + std::fabs(a[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p * 3 + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k * 3 + p], akq = a[k * 3 + q];
          a[k * 3 + p] = c * akp - s * akq;
          a[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
          a[p * 3 + k] = c * apk - s * aqk;
          a[q * 3 + k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = a[0];
  ev[1] = a[4];
  ev[2] = a[8];
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
