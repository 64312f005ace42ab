// pcl/registration/transformation_estimation_svd.h - TransformationEstimationSVD
// (registration/include/pcl/registration/transformation_estimation_svd.h:57-144,
//  impl/transformation_estimation_svd.hpp:43-168) on the device: pclmi_estimate_rigid_svd reduces
// the centroids and the 3x3 cross-covariance in fp64 on the GPU and solves the 3x3 SVD on the host.
#ifndef PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_SVD_H_
#define PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_SVD_H_

#include <pcl/console/print.h>
#include <pcl/registration/transformation_estimation.h>
#include <pclmi.h>

namespace pcl {
namespace registration {

template <typename PointSource, typename PointTarget, typename Scalar = float>
class TransformationEstimationSVD : public TransformationEstimation<PointSource, PointTarget, Scalar> {
 public:
  typedef typename TransformationEstimation<PointSource, PointTarget, Scalar>::Matrix4 Matrix4;
  typedef std::shared_ptr<TransformationEstimationSVD<PointSource, PointTarget, Scalar> > Ptr;

  TransformationEstimationSVD() {}
  virtual ~TransformationEstimationSVD() {}
  int deviceLoop() const override { return 0; }

  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   Matrix4 &transformation_matrix) const override {
    const size_t nr_points = cloud_src.points.size();
    if (cloud_tgt.points.size() != nr_points) {  // impl/transformation_estimation_svd.hpp:50-54
      PCL_ERROR("[pcl::TransformationEstimationSVD::estimateRigidTransformation] Number or points in source (%zu) "
                "differs than target (%zu)!\n", nr_points, cloud_tgt.points.size());
      return;
    }
    run(cloud_src, nullptr, cloud_tgt, nullptr, nr_points, transformation_matrix);
  }

  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src, const std::vector<int> &indices_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   Matrix4 &transformation_matrix) const override {
    if (indices_src.size() != cloud_tgt.points.size()) {  // :69-73
      PCL_ERROR("[pcl::TransformationSVD::estimateRigidTransformation] Number or points in source (%zu) differs "
                "than target (%zu)!\n", indices_src.size(), cloud_tgt.points.size());
      return;
    }
    run(cloud_src, indices_src.data(), cloud_tgt, nullptr, indices_src.size(), transformation_matrix);
  }

  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src, const std::vector<int> &indices_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt, const std::vector<int> &indices_tgt,
                                   Matrix4 &transformation_matrix) const override {
    if (indices_src.size() != indices_tgt.size()) {  // :89-93
      PCL_ERROR("[pcl::TransformationEstimationSVD::estimateRigidTransformation] Number or points in source (%zu) "
                "differs than target (%zu)!\n", indices_src.size(), indices_tgt.size());
      return;
    }
    run(cloud_src, indices_src.data(), cloud_tgt, indices_tgt.data(), indices_src.size(), transformation_matrix);
  }

  void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                   const pcl::PointCloud<PointTarget> &cloud_tgt,
                                   const pcl::Correspondences &correspondences,
                                   Matrix4 &transformation_matrix) const override {
    std::vector<int> si(correspondences.size()), ti(correspondences.size());  // :101-111
    for (size_t i = 0; i < correspondences.size(); ++i) {
      si[i] = correspondences[i].index_query;
      ti[i] = correspondences[i].index_match;
    }
    run(cloud_src, si.data(), cloud_tgt, ti.data(), si.size(), transformation_matrix);
  }

 protected:
  void run(const pcl::PointCloud<PointSource> &src, const int *si, const pcl::PointCloud<PointTarget> &tgt,
           const int *ti, size_t n, Matrix4 &T) const {
    T.setIdentity();
    if (n == 0) return;
    const int rc = pclmi_estimate_rigid_svd(reinterpret_cast<const float *>(src.points.data()), src.points.size(),
                                            sizeof(PointSource), si, reinterpret_cast<const float *>(tgt.points.data()),
                                            tgt.points.size(), sizeof(PointTarget), ti, n, T.data());
    if (rc != PCLMI_OK) PCL_ERROR("[pcl::TransformationEstimationSVD] %s\n", pclmi_last_error());
  }
};

}  // namespace registration
}  // namespace pcl
#endif
