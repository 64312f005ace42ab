// rigid_solve.hpp - the O(1) tail of TransformationEstimationSVD
// (registration/include/pcl/registration/impl/transformation_estimation_svd.hpp:114-168) as
// host+device inline functions: ONE source for pclmi_icp_solve (host) and icp_solve_kernel (device),
// compiled with -ffp-contract=off on both sides and using only IEEE +,-,*,/,sqrt in fp64, so the two
// produce the same bits from the same sums.
//
// The reference does this in fp32 with a two-pass demean; here the sums arrive as one-pass fp64
// moments about a common origin, H = S_st - S_s S_t' / n, and the 3x3 SVD runs in fp64 - at least as
// accurate as the reference, within its 1e-5 tolerance.
#ifndef PCLMI_RIGID_SOLVE_HPP_
#define PCLMI_RIGID_SOLVE_HPP_

#include <hip/hip_runtime.h>

#include "pclmi.h"

namespace pclmi {
namespace rs {

#define PCLMI_HD __host__ __device__ inline

// One rotation of the one-sided Jacobi SVD on columns P, Q (compile-time indices: everything stays in
// registers on the GPU - a run-time index would put the matrices into scratch memory).
template <int P, int Q>
PCLMI_HD bool jacobi_rotate(double *a, double *v) {
  double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    alpha += a[r * 3 + P] * a[r * 3 + P];
    beta += a[r * 3 + Q] * a[r * 3 + Q];
    gamma += a[r * 3 + P] * a[r * 3 + Q];
  }
  // columns orthogonal to a few fp64 roundings: the rounding error of gamma itself is ~eps * sqrt(alpha * beta),
  // a tighter test would keep rotating by noise until the sweep limit
  if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) return false;
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double x = a[r * 3 + P], y = a[r * 3 + Q];
    a[r * 3 + P] = c * x - s * y;
    a[r * 3 + Q] = s * x + c * y;
    x = v[r * 3 + P];
    y = v[r * 3 + Q];
    v[r * 3 + P] = c * x - s * y;
    v[r * 3 + Q] = s * x + c * y;
  }
  return true;
}

// exchange columns I and J of (s, a, v) when s[J] > s[I]
template <int I, int J>
PCLMI_HD void order_columns(double *s, double *a, double *v) {
  if (s[J] > s[I]) {
    double t = s[I];
    s[I] = s[J];
    s[J] = t;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      t = a[r * 3 + I];
      a[r * 3 + I] = a[r * 3 + J];
      a[r * 3 + J] = t;
      t = v[r * 3 + I];
      v[r * 3 + I] = v[r * 3 + J];
      v[r * 3 + J] = t;
    }
  }
}

// one-sided Jacobi SVD of a 3x3, singular values descending (as Eigen::JacobiSVD orders them, so
// that "column 2" of the determinant fix is the smallest singular direction).  A = U diag(S) V'.
// work: 21 doubles of scratch for the rotated matrix, V and the singular values.  On the GPU the caller passes LDS
// (the one lane that solves runs under the step kernel's 64-VGPR budget: ~40 live fp64 values in registers would
// spill to scratch memory on the iteration's critical path; as memory operands they cost LDS latency on this rare
// path only).  Same arithmetic wherever the arrays live.
PCLMI_HD void svd3(const double A[9], double *U, double *S, double *V, double *work) {
  double *a = work, *v = work + 9, *s = work + 18;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    a[i] = A[i];
    v[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = jacobi_rotate<0, 1>(a, v);
    rotated |= jacobi_rotate<0, 2>(a, v);
    rotated |= jacobi_rotate<1, 2>(a, v);
    if (!rotated) break;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) s[c] = sqrt(a[c] * a[c] + a[3 + c] * a[3 + c] + a[6 + c] * a[6 + c]);
  order_columns<0, 1>(s, a, v);
  order_columns<0, 2>(s, a, v);
  order_columns<1, 2>(s, a, v);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    S[c] = s[c];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      V[r * 3 + c] = v[r * 3 + c];
      U[r * 3 + c] = S[c] > 0.0 ? a[r * 3 + c] / S[c] : 0.0;
    }
  }
  // rank-deficient H (three-point RANSAC samples are rank 2): complete U to an orthonormal basis,
  // which is what ComputeFullU returns.
  const double tiny = S[0] * 1e-13;
  if (S[2] <= tiny) {
    double u0[3] = {U[0], U[3], U[6]};
    if (S[0] <= 0.0) {
      u0[0] = 1; u0[1] = 0; u0[2] = 0;
      U[0] = 1; U[3] = 0; U[6] = 0;
    }
    if (S[1] <= tiny) {
      int m = fabs(u0[1]) < fabs(u0[0]) ? 1 : 0;  // the axis u0 is least aligned with
      if (fabs(u0[2]) < fabs(m ? u0[1] : u0[0])) m = 2;
      const double e[3] = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
      double w[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
      const double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
      for (int d = 0; d < 3; ++d) w[d] /= nw;
      U[1] = w[0]; U[4] = w[1]; U[7] = w[2];
    }
    const double u1[3] = {U[1], U[4], U[7]};
    U[2] = u0[1] * u1[2] - u0[2] * u1[1];
    U[5] = u0[2] * u1[0] - u0[0] * u1[2];
    U[8] = u0[0] * u1[1] - u0[1] * u1[0];
  }
}

PCLMI_HD double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// R = V U' of H = U S V' WITHOUT the SVD, for the common case det(H) > 0 and H not close to singular: V U' is
// then the orthogonal polar factor of H', which the scaled Newton iteration X <- (g X + X^-T / g) / 2 (Higham,
// "Computing the polar decomposition - with applications", 1986) reaches in 5-7 steps of one 3x3 cofactor inverse
// each - about 2 us of dependent fp64 latency for one GPU lane against ~10 us for the Jacobi sweeps, and the solve
// sits on the critical path of every ICP iteration.  Backward stable; both give R to a few fp64 ulps of each other,
// far inside the fp32 rounding of T.  Returns false (caller takes the SVD path, which also handles the reflection
// fix of transformation_estimation_svd.hpp:156-160 and rank-deficient H) when the normalised determinant is not
// safely positive.  Only IEEE +,-,*,/,sqrt: same bits on host and device.
//
// What has to be exact and what does not: X <- a X + b cof(X) maps U diag(s) V' to U diag(a s + b det / s) V' for ANY
// scalars a, b - the singular VECTORS (the answer) are untouched, only the path of the singular values to 1 depends
// on them.  So the normalisation of X0 and the scaling factors of the first steps are computed in fp32 (correctly
// rounded / and sqrt on both sides: still the same bits on host and device) - an fp64 division or square root is ~200
// cycles of dependent latency for the one lane that solves, and the scaled steps had five of them each; the plain steps
// at the end (a = 1/2, b = 1/(2 det): the fixed point must be the ORTHOGONAL factor) keep their fp64 division.
PCLMI_HD bool polar_rotation(const double H[9], double R[9]) {
  double f2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) f2 += H[i] * H[i];
  if (!(f2 > 0.0)) return false;
  // ~1 / ||H||_F: a power-of-two-exact fp64 product of the fp32 estimate (any positive factor would do)
  const double inv_f = f2 > 1e-60 && f2 < 1e60 ? (double)(1.0f / sqrtf((float)f2)) : 1.0 / sqrt(f2);
  double X[9];  // X0 = H' / ~||H||_F
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) X[r * 3 + c] = H[c * 3 + r] * inv_f;
  bool scaled = true;
  for (int it = 0; it < 40; ++it) {
    double C[9];  // cofactors: X^-T = C / det
    C[0] = X[4] * X[8] - X[5] * X[7];
    C[1] = X[5] * X[6] - X[3] * X[8];
    C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8];
    C[4] = X[0] * X[8] - X[2] * X[6];
    C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4];
    C[7] = X[2] * X[3] - X[0] * X[5];
    C[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    if (it == 0 && !(det > 1e-9)) return false;  // singular values s1 s2 s3 / ||H||^3: reflection or near-singular
    if (!(det > 0.0)) return false;
    double a, b;  // X <- a X + b C
    bool plain = !scaled;
    if (scaled) {  // g = sqrt(||X^-T||_F / ||X||_F): brings the singular values to 1 from both sides
      double nx = 0.0, nc = 0.0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        nx += X[i] * X[i];
        nc += C[i] * C[i];
      }
      const float detf = (float)det;
      const float g2 = sqrtf((float)nc / (float)nx) / detf;  // g^2
      const float g = sqrtf(g2);
      const float bf = 0.5f / (g * detf);
      // out of fp32's comfortable range (or not finite): the plain step
      if (g > 1e-18f && g < 1e18f && bf > 1e-30f && bf < 1e30f) {
        a = 0.5 * (double)g;
        b = (double)bf;
      } else {
        plain = true;
      }
    }
    if (plain) {
      a = 0.5;
      b = 0.5 / det;
    }
    double d2 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double xn = a * X[i] + b * C[i];
      const double d = xn - X[i];
      d2 += d * d;
      X[i] = xn;
    }
    if (d2 < 1e-4) scaled = false;  // close to orthogonal: the plain iteration converges quadratically from here
    // ||step|| < 1e-10 after a PLAIN step: what is left is ~1e-20, below fp64 resolution (a scaled step's factors are
    // fp32: it may land 1e-7 from orthogonal, and says nothing about the error)
    if (plain && d2 < 1e-20) break;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = X[i];
  return true;
}

// H of (numerical) rank 2 - RANSAC's three-point samples, clouds on one plane - or with a tiny negative determinant
// fails polar_rotation's first test and used to take the Jacobi SVD (~8 us of one lane's fp64 latency, on the critical
// path of every batch of RANSAC trials).  R = V diag(1, 1, det(V U')) U' (the reflection fix of
// transformation_estimation_svd.hpp:156-160 acts on the smallest singular value's pair) is also the polar factor of
//   H + s sigma ns nt'     ns / nt: unit normals of H's column / row space, sigma ~ ||H||, s = +-1 so that det > 0
// because that matrix has the same first two singular pairs and (ns, nt) as the third with a positive determinant.
// ns / nt are the largest cross product of two columns / rows.  Returns false (caller takes the SVD) for rank <= 1.
PCLMI_HD bool rank2_completion(const double H[9], double Hc[9]) {
  double f2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) f2 += H[i] * H[i];
  if (!(f2 > 0.0)) return false;
  // only where the third singular value is noise (|det| = s1 s2 s3 below 1e-12 ||H||^3): a full-rank H with a negative
  // determinant (a genuine reflection) has its own third singular pair, which the cross products below do not give
  if (!(fabs(det3(H)) <= 1e-12 * f2 * sqrt(f2))) return false;
  double ns[3] = {0, 0, 0}, nt[3] = {0, 0, 0}, bs = -1.0, bt = -1.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int b = (a + 1) % 3;
    // columns a, b of H (span of the demeaned source points) and rows a, b (span of the target points)
    const double ca[3] = {H[0 * 3 + a], H[1 * 3 + a], H[2 * 3 + a]}, cb[3] = {H[0 * 3 + b], H[1 * 3 + b], H[2 * 3 + b]};
    const double ra[3] = {H[a * 3 + 0], H[a * 3 + 1], H[a * 3 + 2]}, rb[3] = {H[b * 3 + 0], H[b * 3 + 1], H[b * 3 + 2]};
    const double xs[3] = {ca[1] * cb[2] - ca[2] * cb[1], ca[2] * cb[0] - ca[0] * cb[2], ca[0] * cb[1] - ca[1] * cb[0]};
    const double xt[3] = {ra[1] * rb[2] - ra[2] * rb[1], ra[2] * rb[0] - ra[0] * rb[2], ra[0] * rb[1] - ra[1] * rb[0]};
    const double qs = xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2];
    const double qt = xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2];
    if (qs > bs) {
      bs = qs;
      ns[0] = xs[0]; ns[1] = xs[1]; ns[2] = xs[2];
    }
    if (qt > bt) {
      bt = qt;
      nt[0] = xt[0]; nt[1] = xt[1]; nt[2] = xt[2];
    }
  }
  // |cross| = s1 s2 |sin|: rank <= 1 (or a second singular value below 1e-6 of the first) is left to the SVD
  if (!(bs > 1e-24 * f2 * f2) || !(bt > 1e-24 * f2 * f2)) return false;
  const double is = 1.0 / sqrt(bs), it = 1.0 / sqrt(bt), sigma = sqrt(f2);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Hc[r * 3 + c] = H[r * 3 + c] + sigma * (ns[r] * is) * (nt[c] * it);
  if (!(det3(Hc) > 0.0)) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Hc[r * 3 + c] = H[r * 3 + c] - sigma * (ns[r] * is) * (nt[c] * it);
  }
  return true;
}

// R = V U' through the SVD, with the reflection fix of transformation_estimation_svd.hpp:156-160.
// work: 42 doubles (see svd3).
PCLMI_HD void svd_rotation(const double H[9], double R[9], double *work) {
  double *U = work + 21, *S = work + 30, *V = work + 33;
  svd3(H, U, S, V, work);
  if (det3(U) * det3(V) < 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += V[r * 3 + k] * U[c * 3 + k];  // V * U'
      R[r * 3 + c] = s;
    }
}

// sums layout: pclmi.h (PCLMI_ICP_NSUMS).  Returns false (T = identity) when there is no pair.
PCLMI_HD bool solve_from_sums(const double *sums, float T[16], double *work /* 42 doubles, see svd3 */) {
  const double n = sums[0];
  if (!(n >= 1.0)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    return false;
  }
  // one reciprocal per divisor (an fp64 division is ~200 cycles of one lane's dependent latency, this runs on the
  // critical path of every ICP iteration; the last-bit difference to nine divisions is far inside T's fp32 rounding)
  const double inv_n = 1.0 / n;
  const double inv_c = sums[18] > 0.0 ? 1.0 / sums[18] : 0.0;
  double H[9];
  {
    double ms[3], mt[3];  // centroids relative to the origin
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ms[d] = sums[1 + d] * inv_n;
      mt[d] = sums[4 + d] * inv_n;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) H[r * 3 + c] = sums[7 + r * 3 + c] - n * ms[r] * mt[c];
    // parked in `work` (LDS on the GPU) while the rotation is computed: nothing but the iteration's matrices is live
    // in registers then (the step kernel's budget is 64 VGPRs; what does not fit goes to scratch memory)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      work[33 + d] = ms[d] + sums[19 + d] * inv_c;  // source centroid
      work[36 + d] = mt[d] + sums[19 + d] * inv_c;  // target centroid
    }
  }
  double R[9];
  if (!polar_rotation(H, R)) {
    double Hc[9];
    if (!(rank2_completion(H, Hc) && polar_rotation(Hc, R))) {
      const double keep[6] = {work[33], work[34], work[35], work[36], work[37], work[38]};
      svd_rotation(H, R, work);
#pragma unroll
      for (int d = 0; d < 6; ++d) work[33 + d] = keep[d];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double rc = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) rc += R[r * 3 + c] * work[33 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = (float)R[r * 3 + c];
    T[r * 4 + 3] = (float)(work[36 + r] - rc);  // t = ct - R cs
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
  return true;
}

// sin and cos of x from IEEE +,-,* and floor only (host and device give the same bits; the device maths library's
// fp64 sin / cos would also pull a large argument-reduction routine into the ICP step kernel).  Cody-Waite reduction
// by pi/2 in two pieces, then the Taylor series on [-pi/4, pi/4] up to x^19 / x^18 (truncation < 3e-18): accurate to
// ~1e-16 for |x| < 1e5, far inside the fp32 rounding the results go through.
PCLMI_HD void sincos_small(double x, double *s_out, double *c_out) {
  if (!(fabs(x) <= 1e9)) {  // non-finite or absurd angle (singular geometry): NaN, as garbage in gives garbage out
    *s_out = *c_out = (x - x) / (x - x);
    return;
  }
  const double k = floor(x * 0.63661977236758134308 + 0.5);
  const double r = (x - k * 1.57079632679489655800e+00) - k * 6.12323399573676603587e-17;
  const double z = r * r;
  double s = -8.22063524662432971696e-18;  // -1/19!
  s = s * z + 2.81145725434552076320e-15;  // 1/17!
  s = s * z - 7.64716373181981647590e-13;  // -1/15!
  s = s * z + 1.60590438368216145994e-10;  // 1/13!
  s = s * z - 2.50521083854417187751e-08;  // -1/11!
  s = s * z + 2.75573192239858906526e-06;  // 1/9!
  s = s * z - 1.98412698412698412698e-04;  // -1/7!
  s = s * z + 8.33333333333333333333e-03;  // 1/5!
  s = s * z - 1.66666666666666666667e-01;  // -1/3!
  s = r + r * (z * s);
  double c = 1.56192069685862264622e-16;   // 1/18!
  c = c * z - 4.77947733238738529744e-14;  // -1/16!
  c = c * z + 1.14707455977297247139e-11;  // 1/14!
  c = c * z - 2.08767569878680989792e-09;  // -1/12!
  c = c * z + 2.75573192239858906526e-07;  // 1/10!
  c = c * z - 2.48015873015873015873e-05;  // -1/8!
  c = c * z + 1.38888888888888888889e-03;  // 1/6!
  c = c * z - 4.16666666666666666667e-02;  // -1/4!
  c = c * z + 0.5;
  c = 1.0 - z * c;
  const long long q = (long long)k & 3;  // quadrant
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *s_out = (q & 2) ? -ss : ss;
  *c_out = (q == 1 || q == 2) ? -cc : cc;
}

// TransformationEstimationPointToPlaneLLS: x = ATA^-1 ATb (Gauss-Jordan with partial pivoting in fp64, standing in
// for Eigen's ATA.inverse () * ATb) and constructTransformationMatrix
// (registration/include/pcl/registration/impl/transformation_estimation_point_to_plane_lls.hpp:118-139,230-245).
// sums: the 21 upper-triangle entries of ATA row by row, then the 6 of ATb.  work: 42 doubles (the augmented 6x7
// system; LDS on the GPU: the elimination indexes it with run-time indices).  Returns false (T = identity) when the
// normal equations are singular.
PCLMI_HD bool solve_point_to_plane_sums(const double *sums, float T[16], double *work) {
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  double *M = work;  // [6][7]
  int o = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      const double v = sums[o++];
      M[r * 7 + c] = v;
      M[c * 7 + r] = v;
    }
  for (int r = 0; r < 6; ++r) M[r * 7 + 6] = sums[21 + r];
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(M[r * 7 + c]) > fabs(M[piv * 7 + c])) piv = r;
    const double d = M[piv * 7 + c];
    if (d == 0.0 || !(fabs(d) <= 1.7976931348623157e308)) return false;
    if (piv != c)
      for (int k = 0; k < 7; ++k) {
        const double t = M[c * 7 + k];
        M[c * 7 + k] = M[piv * 7 + k];
        M[piv * 7 + k] = t;
      }
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = M[r * 7 + c] / M[c * 7 + c];
      for (int k = c; k < 7; ++k) M[r * 7 + k] -= f * M[c * 7 + k];
    }
  }
  double sa, ca, sb, cb, sg, cg;
  sincos_small(M[0 * 7 + 6] / M[0 * 7 + 0], &sa, &ca);  // alpha
  sincos_small(M[1 * 7 + 6] / M[1 * 7 + 1], &sb, &cb);  // beta
  sincos_small(M[2 * 7 + 6] / M[2 * 7 + 2], &sg, &cg);  // gamma
  T[0] = (float)(cg * cb);
  T[1] = -(float)(sg * ca + cg * sb * sa);
  T[2] = (float)(sg * sa + cg * sb * ca);
  T[4] = (float)(sg * cb);
  T[5] = (float)(cg * ca + sg * sb * sa);
  T[6] = -(float)(cg * sa + sg * sb * ca);
  T[8] = -(float)(sb);
  T[9] = (float)(cb * sa);
  T[10] = (float)(cb * ca);
  T[3] = (float)(M[3 * 7 + 6] / M[3 * 7 + 3]);
  T[7] = (float)(M[4 * 7 + 6] / M[4 * 7 + 4]);
  T[11] = (float)(M[5 * 7 + 6] / M[5 * 7 + 5]);
  return true;
}

// C = A * B (row-major 4x4, fp32, k ascending); C may alias A or B
PCLMI_HD void mat4_mul(const float A[16], const float B[16], float C[16]) {
  float R[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      R[r * 4 + c] = s;
    }
#pragma unroll
  for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// sum |T - Tprev| in fp32 (the Eigen expression of icp.hpp:194)
PCLMI_HD float transform_delta(const float T[16], const float Tprev[16]) {
  float fs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) fs += fabsf(T[i] - Tprev[i]);
  return fs;
}

}  // namespace rs
}  // namespace pclmi
#endif
