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

}  // namespace pcl
#endif
