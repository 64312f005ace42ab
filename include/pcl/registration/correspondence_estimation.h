// pcl/registration/correspondence_estimation.h - CorrespondenceEstimation
// (registration/include/pcl/registration/correspondence_estimation.h:77-220,
//  impl/correspondence_estimation.hpp:47-233): batched 1-NN of the source in the target tree on the
// device.  The target tree is built once per setInputTarget (the reference rebuilds it on every
// determine* call, impl/correspondence_estimation.hpp:47-63 - same results).
#ifndef PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_H_
#define PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_H_

#include <limits>
#include <string>

#include <pcl/correspondence.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/pcl_base.h>

namespace pcl {
namespace registration {

template <typename PointSource, typename PointTarget>
class CorrespondenceEstimation : public PCLBase<PointSource> {
 public:
  using PCLBase<PointSource>::initCompute;
  using PCLBase<PointSource>::deinitCompute;
  using PCLBase<PointSource>::input_;
  using PCLBase<PointSource>::indices_;
  using PCLBase<PointSource>::fake_indices_;

  typedef pcl::KdTreeFLANN<PointTarget> KdTree;
  typedef std::shared_ptr<KdTree> KdTreePtr;
  typedef pcl::PointCloud<PointSource> PointCloudSource;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef pcl::PointCloud<PointTarget> PointCloudTarget;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef std::shared_ptr<CorrespondenceEstimation<PointSource, PointTarget> > Ptr;

  CorrespondenceEstimation() : corr_name_("CorrespondenceEstimation"), tree_(new KdTree), target_dirty_(false) {}
  virtual ~CorrespondenceEstimation() {}

  virtual void setInputTarget(const PointCloudTargetConstPtr &cloud) {
    if (!cloud || cloud->points.empty()) {
      PCL_ERROR("[pcl::%s::setInputTarget] Invalid or empty point cloud dataset given!\n", getClassName().c_str());
      return;
    }
    target_ = cloud;
    target_dirty_ = true;
  }
  PointCloudTargetConstPtr const getInputTarget() { return target_; }

  // impl/correspondence_estimation.hpp:83-141; keeps pairs with d2 <= max_distance^2
  virtual void determineCorrespondences(pcl::Correspondences &correspondences,
                                        double max_distance = std::numeric_limits<double>::max()) {
    correspondences.clear();
    if (!initComputeTarget()) return;
    const size_t n = fake_indices_ ? input_->points.size() : indices_->size();
    correspondences.resize(n);
    size_t n_out = 0;
    const int rc = pclmi_correspondences(
        tree_->getDeviceTree(), reinterpret_cast<const float *>(input_->points.data()), input_->points.size(),
        sizeof(PointSource), fake_indices_ ? nullptr : reinterpret_cast<const int32_t *>(indices_->data()),
        fake_indices_ ? 0 : indices_->size(), max_distance, reinterpret_cast<pclmi_correspondence *>(correspondences.data()),
        &n_out);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::%s::determineCorrespondences] %s\n", getClassName().c_str(), pclmi_last_error());
      n_out = 0;
    }
    correspondences.resize(n_out);
    deinitCompute();
  }

  // impl/correspondence_estimation.hpp:144-233
  virtual void determineReciprocalCorrespondences(pcl::Correspondences &correspondences,
                                                  double max_distance = std::numeric_limits<double>::max()) {
    correspondences.clear();
    if (!initComputeTarget()) return;
    pcl::KdTreeFLANN<PointSource> tree_reciprocal;
    tree_reciprocal.setInputCloud(input_, fake_indices_ ? IndicesConstPtr() : IndicesConstPtr(indices_));
    if (!tree_reciprocal.getDeviceTree()) return;
    const size_t n = fake_indices_ ? input_->points.size() : indices_->size();
    correspondences.resize(n);
    size_t n_out = 0;
    const int rc = pclmi_correspondences_reciprocal(
        tree_->getDeviceTree(), tree_reciprocal.getDeviceTree(), reinterpret_cast<const float *>(input_->points.data()),
        input_->points.size(), sizeof(PointSource),
        fake_indices_ ? nullptr : reinterpret_cast<const int32_t *>(indices_->data()),
        fake_indices_ ? 0 : indices_->size(), reinterpret_cast<const float *>(target_->points.data()),
        target_->points.size(), sizeof(PointTarget), max_distance,
        reinterpret_cast<pclmi_correspondence *>(correspondences.data()), &n_out);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::%s::determineReciprocalCorrespondences] %s\n", getClassName().c_str(), pclmi_last_error());
      n_out = 0;
    }
    correspondences.resize(n_out);
    deinitCompute();
  }

 protected:
  bool initComputeTarget() {
    if (!target_) {
      PCL_WARN("[pcl::%s::compute] No input target dataset was given!\n", getClassName().c_str());
      return false;
    }
    if (target_dirty_) {
      tree_->setInputCloud(target_);
      target_dirty_ = false;
    }
    if (!tree_->getDeviceTree()) return false;
    return initCompute();
  }
  const std::string &getClassName() const { return corr_name_; }

  std::string corr_name_;
  KdTreePtr tree_;
  PointCloudTargetConstPtr target_;
  bool target_dirty_;
};

}  // namespace registration
}  // namespace pcl
#endif
