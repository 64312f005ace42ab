// pcl/point_types.h - the point structs the kNN / ICP path touches.
// Layout follows common/include/pcl/impl/point_types.hpp:141-155,211-237: 16-byte aligned
// `float data[4]` with x, y, z first and data[3] = 1.  Any struct whose first three floats are
// x, y, z works with the facade templates (the library reads 3 floats at a byte stride).
#ifndef PCLMI_FACADE_POINT_TYPES_H_
#define PCLMI_FACADE_POINT_TYPES_H_

#include <cstdint>
#include <ostream>

namespace pcl {

#define PCLMI_ADD_POINT4D   \
  union alignas(16) {       \
    float data[4];          \
    struct {                \
      float x, y, z;        \
    };                      \
  };

struct PointXYZ {
  PCLMI_ADD_POINT4D
  PointXYZ() : data{0.f, 0.f, 0.f, 1.f} {}
  PointXYZ(float x_, float y_, float z_) : data{x_, y_, z_, 1.f} {}
};

struct PointXYZI {
  PCLMI_ADD_POINT4D
  union alignas(16) {
    float data_c[4];
    struct {
      float intensity;
    };
  };
  PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
  PointXYZI(float x_, float y_, float z_, float i_ = 0.f) : data{x_, y_, z_, 1.f}, data_c{i_, 0.f, 0.f, 0.f} {}
};

struct PointNormal {
  PCLMI_ADD_POINT4D
  union alignas(16) {
    float data_n[4];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union alignas(16) {
    float data_c[4];
    struct {
      float curvature;
    };
  };
  PointNormal() : data{0.f, 0.f, 0.f, 1.f}, data_n{0.f, 0.f, 0.f, 0.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};

// common/include/pcl/impl/point_types.hpp: Normal = normal[3] (+pad) and curvature (+pad), 32 bytes
struct Normal {
  union alignas(16) {
    float data_n[4];
    float normal[3];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union alignas(16) {
    float data_c[4];
    struct {
      float curvature;
    };
  };
  Normal() : data_n{0.f, 0.f, 0.f, 0.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
  Normal(float nx, float ny, float nz) : data_n{nx, ny, nz, 0.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};

inline std::ostream &operator<<(std::ostream &os, const PointXYZ &p) {
  return os << "(" << p.x << "," << p.y << "," << p.z << ")";
}

static_assert(sizeof(PointXYZ) == 16, "PointXYZ must be 16 bytes");

}  // namespace pcl
#endif
