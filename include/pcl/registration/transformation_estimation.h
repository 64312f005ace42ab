// pcl/registration/transformation_estimation.h - abstract TransformationEstimation
// (registration/include/pcl/registration/transformation_estimation.h).
#ifndef PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_H_
#define PCLMI_FACADE_REGISTRATION_TRANSFORMATION_ESTIMATION_H_

#include <memory>
#include <vector>

#include <pcl/common/matrix4.h>
#include <pcl/correspondence.h>
#include <pcl/point_cloud.h>

namespace pcl {
namespace registration {

// Scalar is kept for signature compatibility; the 4x4 handed back is fp32 (the sums behind it are
// reduced in fp64 on the device either way).
template <typename PointSource, typename PointTarget, typename Scalar = float>
class TransformationEstimation {
 public:
  typedef pcl::Matrix4f Matrix4;
  typedef std::shared_ptr<TransformationEstimation<PointSource, PointTarget, Scalar> > Ptr;
  typedef std::shared_ptr<const TransformationEstimation<PointSource, PointTarget, Scalar> > ConstPtr;

  TransformationEstimation() {}
  virtual ~TransformationEstimation() {}

  /** \brief Not in the reference: which fused device loop IterativeClosestPoint runs for this estimator
    * (0: the SVD loop, 1: the point-to-plane loop of pclmi_icp_align_point_to_plane, -1: none - align() reports an
    * error: the per-iteration estimator call of icp.hpp:171 does not leave the device). */
  virtual int deviceLoop() const { return -1; }

  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                           const pcl::PointCloud<PointTarget> &cloud_tgt,
                                           Matrix4 &transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                           const std::vector<int> &indices_src,
                                           const pcl::PointCloud<PointTarget> &cloud_tgt,
                                           Matrix4 &transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                           const std::vector<int> &indices_src,
                                           const pcl::PointCloud<PointTarget> &cloud_tgt,
                                           const std::vector<int> &indices_tgt,
                                           Matrix4 &transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource> &cloud_src,
                                           const pcl::PointCloud<PointTarget> &cloud_tgt,
                                           const pcl::Correspondences &correspondences,
                                           Matrix4 &transformation_matrix) const = 0;
};

}  // namespace registration
}  // namespace pcl
#endif
