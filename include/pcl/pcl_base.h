// pcl/pcl_base.h - PCLBase<PointT>: input cloud + optional index view
// (common/include/pcl/pcl_base.h:73-253).
#ifndef PCLMI_FACADE_PCL_BASE_H_
#define PCLMI_FACADE_PCL_BASE_H_

#include <pcl/point_cloud.h>

namespace pcl {

template <typename PointT>
class PCLBase {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef typename PointCloud::ConstPtr PointCloudConstPtr;

  PCLBase() : use_indices_(false), fake_indices_(false) {}
  virtual ~PCLBase() {}

  virtual void setInputCloud(const PointCloudConstPtr &cloud) { input_ = cloud; }
  PointCloudConstPtr const getInputCloud() { return input_; }
  void setIndices(const IndicesPtr &indices) {
    indices_ = indices;
    fake_indices_ = false;
    use_indices_ = true;
  }
  void setIndices(const IndicesConstPtr &indices) {
    indices_.reset(new std::vector<int>(*indices));
    fake_indices_ = false;
    use_indices_ = true;
  }
  IndicesPtr const getIndices() { return indices_; }
  const PointT &operator[](size_t pos) { return (*input_)[(*indices_)[pos]]; }

 protected:
  // pcl_base.h:221-253: without user indices, an identity list over the whole cloud
  bool initCompute() {
    if (!input_) return false;
    if (!indices_) {
      fake_indices_ = true;
      indices_.reset(new std::vector<int>);
    }
    if (fake_indices_ && indices_->size() != input_->points.size()) {
      const size_t old = indices_->size();
      indices_->resize(input_->points.size());
      for (size_t i = old; i < indices_->size(); ++i) (*indices_)[i] = static_cast<int>(i);
      if (old > indices_->size()) indices_->resize(input_->points.size());
    }
    return true;
  }
  bool deinitCompute() { return true; }

  PointCloudConstPtr input_;
  IndicesPtr indices_;
  bool use_indices_;
  bool fake_indices_;
};

}  // namespace pcl
#endif
