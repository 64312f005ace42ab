// pcl/common/io.h - pcl::copyPointCloud between point types (common/include/pcl/common/io.h,
// impl/io.hpp copyPointCloud): every field the two types share is copied, the rest of the output point keeps its
// default value.  The facade's point types share x, y, z and - when both have them - normal_x/y/z and curvature.
#ifndef PCLMI_FACADE_COMMON_IO_H_
#define PCLMI_FACADE_COMMON_IO_H_

#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

namespace pcl {
namespace detail {
template <typename A, typename B>
auto copy_normal_fields(const A &a, B &b, int) -> decltype((void)a.normal_x, (void)b.normal_x, void()) {
  b.normal_x = a.normal_x;
  b.normal_y = a.normal_y;
  b.normal_z = a.normal_z;
  b.curvature = a.curvature;
}
template <typename A, typename B>
void copy_normal_fields(const A &, B &, long) {}
template <typename A, typename B>
auto copy_xyz_fields(const A &a, B &b, int) -> decltype((void)a.x, (void)b.x, void()) {
  b.x = a.x;
  b.y = a.y;
  b.z = a.z;
}
template <typename A, typename B>
void copy_xyz_fields(const A &, B &, long) {}
}  // namespace detail

template <typename PointInT, typename PointOutT>
void copyPointCloud(const pcl::PointCloud<PointInT> &cloud_in, pcl::PointCloud<PointOutT> &cloud_out) {
  cloud_out.header = cloud_in.header;
  cloud_out.width = cloud_in.width;
  cloud_out.height = cloud_in.height;
  cloud_out.is_dense = cloud_in.is_dense;
  cloud_out.points.assign(cloud_in.points.size(), PointOutT());
  for (size_t i = 0; i < cloud_in.points.size(); ++i) {
    detail::copy_xyz_fields(cloud_in.points[i], cloud_out.points[i], 0);
    detail::copy_normal_fields(cloud_in.points[i], cloud_out.points[i], 0);
  }
}

// the index-list overload (io.hpp): output = input[indices], one row
template <typename PointInT, typename PointOutT>
void copyPointCloud(const pcl::PointCloud<PointInT> &cloud_in, const std::vector<int> &indices,
                    pcl::PointCloud<PointOutT> &cloud_out) {
  cloud_out.header = cloud_in.header;
  cloud_out.width = static_cast<uint32_t>(indices.size());
  cloud_out.height = 1;
  cloud_out.is_dense = cloud_in.is_dense;
  cloud_out.points.assign(indices.size(), PointOutT());
  for (size_t i = 0; i < indices.size(); ++i) {
    detail::copy_xyz_fields(cloud_in.points[indices[i]], cloud_out.points[i], 0);
    detail::copy_normal_fields(cloud_in.points[indices[i]], cloud_out.points[i], 0);
  }
}

}  // namespace pcl
#endif
