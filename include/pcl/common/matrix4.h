// pcl/common/matrix4.h - the minimal 4x4 / 4x1 float types of the registration API.
// The reference's API uses Eigen::Matrix4f / Eigen::Vector4f; Eigen is not a dependency of this
// facade (and is absent from the build image), so these stand in with the subset of the interface
// the registration callers use: Identity(), setIdentity(), operator()(r, c), operator*, block
// access through rotation() / translation(), data() (ROW-major, what pclmi.h takes).
#ifndef PCLMI_FACADE_COMMON_MATRIX4_H_
#define PCLMI_FACADE_COMMON_MATRIX4_H_

#include <cmath>
#include <cstring>
#include <ostream>

namespace pcl {

struct Vector4f {
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  Vector4f() {}
  Vector4f(float a, float b, float c, float d) : v{a, b, c, d} {}
  float &operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  float &operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
};

struct Matrix4f {
  float m[16];
  Matrix4f() { setIdentity(); }
  static Matrix4f Identity() { return Matrix4f(); }
  void setIdentity() {
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
  }
  bool isIdentity() const {
    for (int i = 0; i < 16; ++i)
      if (m[i] != ((i % 5 == 0) ? 1.f : 0.f)) return false;
    return true;
  }
  float &operator()(int r, int c) { return m[r * 4 + c]; }
  float operator()(int r, int c) const { return m[r * 4 + c]; }
  float *data() { return m; }  // row-major
  const float *data() const { return m; }
  Matrix4f operator*(const Matrix4f &b) const {
    Matrix4f r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += m[i * 4 + k] * b.m[k * 4 + j];
        r.m[i * 4 + j] = s;
      }
    return r;
  }
  Vector4f operator*(const Vector4f &x) const {
    Vector4f r;
    for (int i = 0; i < 4; ++i) r.v[i] = ((m[i * 4] * x.v[0] + m[i * 4 + 1] * x.v[1]) + m[i * 4 + 2] * x.v[2]) + m[i * 4 + 3] * x.v[3];
    return r;
  }
  bool operator==(const Matrix4f &o) const { return std::memcmp(m, o.m, sizeof(m)) == 0; }
  bool operator!=(const Matrix4f &o) const { return !(*this == o); }
  // sum |a - b| over all coefficients: (T - T_prev).array().abs().sum() of icp.hpp:194
  float absDiffSum(const Matrix4f &o) const {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += std::fabs(m[i] - o.m[i]);
    return s;
  }
};

inline std::ostream &operator<<(std::ostream &os, const Matrix4f &M) {
  for (int r = 0; r < 4; ++r) os << M(r, 0) << " " << M(r, 1) << " " << M(r, 2) << " " << M(r, 3) << "\n";
  return os;
}

}  // namespace pcl
#endif
