// pcl/common/transforms.h - transformPointCloud (common/include/pcl/common/impl/transforms.hpp:168-206)
// on the device: p <- R p + t in fp32, non-finite points left untouched.
#ifndef PCLMI_FACADE_COMMON_TRANSFORMS_H_
#define PCLMI_FACADE_COMMON_TRANSFORMS_H_

#include <pcl/common/matrix4.h>
#include <pcl/console/print.h>
#include <pcl/point_cloud.h>
#include <pclmi.h>

namespace pcl {

template <typename PointT>
void transformPointCloud(const pcl::PointCloud<PointT> &cloud_in, pcl::PointCloud<PointT> &cloud_out,
                         const Matrix4f &transform) {
  if (&cloud_in != &cloud_out) {  // transforms.hpp:172-181
    cloud_out.header = cloud_in.header;
    cloud_out.is_dense = cloud_in.is_dense;
    cloud_out.width = cloud_in.width;
    cloud_out.height = cloud_in.height;
    cloud_out.points = cloud_in.points;
  }
  if (cloud_out.points.empty()) return;
  const int rc = pclmi_transform(reinterpret_cast<float *>(&cloud_out.points[0]), cloud_out.points.size(),
                                 sizeof(PointT), transform.data());
  if (rc != PCLMI_OK) PCL_ERROR("[pcl::transformPointCloud] %s\n", pclmi_last_error());
}

/** \brief transformPointCloudWithNormals (common/include/pcl/common/impl/transforms.hpp): p <- R p + t and
 *  n <- R n in fp32.  Host code (the device path is transformPointCloud; this one is used to prepare test data). */
template <typename PointT>
void transformPointCloudWithNormals(const pcl::PointCloud<PointT> &cloud_in, pcl::PointCloud<PointT> &cloud_out,
                                    const Matrix4f &T) {
  if (&cloud_in != &cloud_out) cloud_out = cloud_in;
  for (size_t i = 0; i < cloud_out.points.size(); ++i) {
    PointT &p = cloud_out.points[i];
    const float x = p.x, y = p.y, z = p.z, nx = p.normal_x, ny = p.normal_y, nz = p.normal_z;
    p.x = ((T(0, 0) * x + T(0, 1) * y) + T(0, 2) * z) + T(0, 3);
    p.y = ((T(1, 0) * x + T(1, 1) * y) + T(1, 2) * z) + T(1, 3);
    p.z = ((T(2, 0) * x + T(2, 1) * y) + T(2, 2) * z) + T(2, 3);
    p.normal_x = (T(0, 0) * nx + T(0, 1) * ny) + T(0, 2) * nz;
    p.normal_y = (T(1, 0) * nx + T(1, 1) * ny) + T(1, 2) * nz;
    p.normal_z = (T(2, 0) * nx + T(2, 1) * ny) + T(2, 2) * nz;
  }
}

}  // namespace pcl
#endif
