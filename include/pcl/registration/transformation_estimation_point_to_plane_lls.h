// pcl/registration/transformation_estimation_point_to_plane_lls.h - TransformationEstimationPointToPlaneLLS
// (registration/include/pcl/registration/transformation_estimation_point_to_plane_lls.h,
//  impl/transformation_estimation_point_to_plane_lls.hpp:45-245): linearised point-to-plane least squares.
// The 6x6 normal equations are reduced on the device (pclmi_estimate_point_to_plane_lls), solved on the host.
// PointSource and PointTarget need x, y, z and normal_x, normal_y, normal_z (e.g. pcl::PointNormal).
#ifndef PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_POINT_TO_PLANE_LLS_H_
#define PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_POINT_TO_PLANE_LLS_H_

#include <pcl/console/print.h>
#include <pcl/registration/transformation_estimation.h>
#include <pclmi.h>

namespace pcl {
namespace registration {

template <typename PointSource, typename PointTarget, typename Scalar = float>
class TransformationEstimationPointToPlaneLLS : public TransformationEstimation<PointSource, PointTarget, Scalar> {
 public:
  typedef std::shared_ptr<TransformationEstimationPointToPlaneLLS<PointSource, PointTarget, Scalar> > Ptr;
  typedef typename TransformationEstimation<PointSource, PointTarget, Scalar>::Matrix4 Matrix4;

  TransformationEstimationPointToPlaneLLS() {}
  virtual ~TransformationEstimationPointToPlaneLLS() {}
  int deviceLoop() const override { return 1; }

  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   Matrix4 &transformation_matrix) const override {
    if (cloud_tgt.points.size() != cloud_src.points.size()) {  // :50-55
      PCL_ERROR("[pcl::TransformationEstimationPointToPlaneLLS::estimateRigidTransformation] Number or points in source "
                "(%zu) differs than target (%zu)!\n", cloud_src.points.size(), cloud_tgt.points.size());
      return;
    }
    run(cloud_src, nullptr, cloud_tgt, nullptr, cloud_src.points.size(), transformation_matrix);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src, const std::vector<int> &indices_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   Matrix4 &transformation_matrix) const override {
    if (cloud_tgt.points.size() != indices_src.size()) {  // :69-74
      PCL_ERROR("[pcl::TransformationEstimationPointToPlaneLLS::estimateRigidTransformation] Number or points in source "
                "(%zu) differs than target (%zu)!\n", indices_src.size(), cloud_tgt.points.size());
      return;
    }
    run(cloud_src, indices_src.data(), cloud_tgt, nullptr, indices_src.size(), transformation_matrix);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src, const std::vector<int> &indices_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt, const std::vector<int> &indices_tgt,
                                   Matrix4 &transformation_matrix) const override {
    if (indices_tgt.size() != indices_src.size()) {  // :90-95
      PCL_ERROR("[pcl::TransformationEstimationPointToPlaneLLS::estimateRigidTransformation] Number or points in source "
                "(%zu) differs than target (%zu)!\n", indices_src.size(), indices_tgt.size());
      return;
    }
    run(cloud_src, indices_src.data(), cloud_tgt, indices_tgt.data(), indices_src.size(), transformation_matrix);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   const pcl::Correspondences &correspondences,
                                   Matrix4 &transformation_matrix) const override {
    std::vector<int> is(correspondences.size()), it(correspondences.size());
    for (size_t i = 0; i < correspondences.size(); ++i) {
      is[i] = correspondences[i].index_query;
      it[i] = correspondences[i].index_match;
    }
    run(cloud_src, is.data(), cloud_tgt, it.data(), is.size(), transformation_matrix);
  }

 protected:
  void run(const pcl::PointCloud<PointSource> &src, const int *is, const pcl::PointCloud<PointTarget> &tgt, const int *it,
           size_t n, Matrix4 &T) const {
    T.setIdentity();
    if (n == 0 || src.points.empty() || tgt.points.empty()) return;
    const int rc = pclmi_estimate_point_to_plane_lls(
        &src.points[0].x, src.points.size(), sizeof(PointSource), &src.points[0].normal_x, sizeof(PointSource), is,
        &tgt.points[0].x, tgt.points.size(), sizeof(PointTarget), &tgt.points[0].normal_x, sizeof(PointTarget), it, n,
        T.data());
    if (rc != PCLMI_OK)
      PCL_ERROR("[pcl::TransformationEstimationPointToPlaneLLS::estimateRigidTransformation] %s\n", pclmi_last_error());
  }
};

}  // namespace registration
}  // namespace pcl
#endif
