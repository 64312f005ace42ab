// pcl/registration/correspondence_rejection.h - CorrespondenceRejector base and the rejectors of the
// ICP path (registration/include/pcl/registration/correspondence_rejection*.h), each one call into
// the device library (pclmi_reject_*).
#ifndef PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_REJECTION_H_
#define PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_REJECTION_H_

#include <algorithm>
#include <limits>
#include <string>

#include <pcl/common/matrix4.h>
#include <pcl/console/print.h>
#include <pcl/correspondence.h>
#include <pcl/point_cloud.h>
#include <pclmi.h>

namespace pcl {

// correspondence.h: query indices present in `before` but not in `after`
inline void getRejectedQueryIndices(const pcl::Correspondences &before, const pcl::Correspondences &after,
                                    std::vector<int> &indices, bool presorting_required = true) {
  indices.clear();
  std::vector<int> a(before.size()), b(after.size());
  for (size_t i = 0; i < before.size(); ++i) a[i] = before[i].index_query;
  for (size_t i = 0; i < after.size(); ++i) b[i] = after[i].index_query;
  if (presorting_required) {
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
  }
  std::set_difference(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(indices));
}

namespace registration {

// correspondence_rejection.h:56-142
class CorrespondenceRejector {
 public:
  typedef std::shared_ptr<CorrespondenceRejector> Ptr;
  CorrespondenceRejector() {}
  virtual ~CorrespondenceRejector() {}

  virtual void setInputCorrespondences(const CorrespondencesConstPtr &correspondences) {
    input_correspondences_ = correspondences;
  }
  CorrespondencesConstPtr getInputCorrespondences() { return input_correspondences_; }

  void getCorrespondences(pcl::Correspondences &correspondences) {  // :86-93
    if (!input_correspondences_ || input_correspondences_->empty()) return;
    applyRejection(correspondences);
  }

  virtual void getRemainingCorrespondences(const pcl::Correspondences &original_correspondences,
                                           pcl::Correspondences &remaining_correspondences) = 0;

  void getRejectedQueryIndices(const pcl::Correspondences &correspondences, std::vector<int> &indices) {  // :113-125
    if (!input_correspondences_ || input_correspondences_->empty()) {
      PCL_WARN("[pcl::%s::getRejectedQueryIndices] Input correspondences not set (lookup of rejected "
               "correspondences _not_ possible).\n", getClassName().c_str());
      return;
    }
    pcl::getRejectedQueryIndices(*input_correspondences_, correspondences, indices);
  }

 protected:
  virtual void applyRejection(pcl::Correspondences &correspondences) {
    getRemainingCorrespondences(*input_correspondences_, correspondences);
  }
  const std::string &getClassName() const { return rejection_name_; }
  static const pclmi_correspondence *in(const pcl::Correspondences &c) {
    return reinterpret_cast<const pclmi_correspondence *>(c.data());
  }
  static pclmi_correspondence *out(pcl::Correspondences &c) { return reinterpret_cast<pclmi_correspondence *>(c.data()); }
  void finish(int rc, size_t n_out, pcl::Correspondences &remaining) const {
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::registration::%s] %s\n", rejection_name_.c_str(), pclmi_last_error());
      n_out = 0;
    }
    remaining.resize(n_out);
  }

  std::string rejection_name_;
  CorrespondencesConstPtr input_correspondences_;
};

// correspondence_rejection_distance.h / impl/_distance.hpp:42-69
class CorrespondenceRejectorDistance : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorDistance() : max_distance_(std::numeric_limits<float>::max()) {
    rejection_name_ = "CorrespondenceRejectorDistance";
  }
  // stores the square, as the reference (correspondence_rejection_distance.h:88); the getter returns it
  virtual void setMaximumDistance(float distance) {
    max_distance_ = distance * distance;
    max_distance_arg_ = distance;
  }
  float getMaximumDistance() { return max_distance_; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_distance(in(original), original.size(), max_distance_arg_, out(tmp), &n_out);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  float max_distance_;
  float max_distance_arg_ = std::sqrt(std::numeric_limits<float>::max());
};

// correspondence_rejection_median_distance.h / impl/_median_distance.hpp:44-69
class CorrespondenceRejectorMedianDistance : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorMedianDistance() : median_distance_(0), factor_(1.0) {
    rejection_name_ = "CorrespondenceRejectorMedianDistance";
  }
  double getMedianDistance() const { return median_distance_; }
  void setMedianFactor(double factor) { factor_ = factor; }
  double getMedianFactor() const { return factor_; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_median(in(original), original.size(), factor_, out(tmp), &n_out, &median_distance_);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  double median_distance_;
  double factor_;
};

// correspondence_rejection_one_to_one.h / impl/_one_to_one.hpp:42-67
class CorrespondenceRejectorOneToOne : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorOneToOne() { rejection_name_ = "CorrespondenceRejectorOneToOne"; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_one_to_one(in(original), original.size(), out(tmp), &n_out);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }
};

// correspondence_rejection_trimmed.h / impl/_trimmed.hpp:42-58
class CorrespondenceRejectorTrimmed : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorTrimmed() : overlap_ratio_(0.5f), nr_min_correspondences_(0) {
    rejection_name_ = "CorrespondenceRejectorTrimmed";
  }
  virtual void setOverlapRadio(float ratio) { overlap_ratio_ = std::min(1.0f, std::max(0.0f, ratio)); }  // [sic]
  float getOverlapRadio() { return overlap_ratio_; }
  void setMinCorrespondences(unsigned int min_correspondences) { nr_min_correspondences_ = min_correspondences; }
  unsigned int getMinCorrespondences() { return nr_min_correspondences_; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_trimmed(in(original), original.size(), overlap_ratio_, nr_min_correspondences_, out(tmp),
                                        &n_out);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  float overlap_ratio_;
  unsigned int nr_min_correspondences_;
};

// correspondence_rejection_var_trimmed.h / impl/_var_trimmed.hpp:44-107
class CorrespondenceRejectorVarTrimmed : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorVarTrimmed() : trimmed_distance_(0), factor_(0), min_ratio_(0.05), max_ratio_(0.95), lambda_(0.95) {
    rejection_name_ = "CorrespondenceRejectorVarTrimmed";
  }
  double getTrimmedDistance() const { return trimmed_distance_; }
  double getTrimFactor() const { return factor_; }
  void setMinRatio(double ratio) { min_ratio_ = ratio; }
  double getMinRatio() const { return min_ratio_; }
  void setMaxRatio(double ratio) { max_ratio_ = ratio; }
  double getMaxRatio() const { return max_ratio_; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    // by_distance_ == false: the reference as shipped (its loop reads the sorted scratch array, so the first
    // #(sorted < trimmed) inputs are kept); setSelectByOwnDistance(true) judges every pair by its own distance
    const int rc = pclmi_reject_var_trimmed(in(original), original.size(), min_ratio_, max_ratio_, lambda_,
                                            by_distance_ ? 1 : 0, out(tmp), &n_out, &factor_, &trimmed_distance_);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  double trimmed_distance_, factor_, min_ratio_, max_ratio_, lambda_;
  bool by_distance_ = false;

 public:
  /** \brief Not in the reference: select by each correspondence's own distance (the documented intent) instead of
    * reproducing the reference's as-shipped selection (correspondence_rejection_var_trimmed.hpp:69-76). */
  void setSelectByOwnDistance(bool on) { by_distance_ = on; }
};

// correspondence_rejection_sample_consensus.h / impl/_sample_consensus.hpp:44-110
template <typename PointT>
class CorrespondenceRejectorSampleConsensus : public CorrespondenceRejector {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::ConstPtr PointCloudConstPtr;
  CorrespondenceRejectorSampleConsensus() : inlier_threshold_(0.05), max_iterations_(std::numeric_limits<int>::max()) {
    rejection_name_ = "CorrespondenceRejectorSampleConsensus";
  }
  virtual void setInputCloud(const PointCloudConstPtr &cloud) { input_ = cloud; }
  virtual void setTargetCloud(const PointCloudConstPtr &cloud) { target_ = cloud; }
  void setInlierThreshold(double threshold) { inlier_threshold_ = threshold; }
  double getInlierThreshold() { return inlier_threshold_; }
  void setMaxIterations(int max_iterations) { max_iterations_ = std::max(max_iterations, 0); }
  int getMaxIterations() { return max_iterations_; }
  Matrix4f getBestTransformation() { return best_transformation_; }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    if (!input_ || !target_) {
      PCL_ERROR("[pcl::registration::%s] input / target cloud not set\n", rejection_name_.c_str());
      remaining = original;
      return;
    }
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_sample_consensus(
        reinterpret_cast<const float *>(input_->points.data()), input_->points.size(), sizeof(PointT),
        reinterpret_cast<const float *>(target_->points.data()), target_->points.size(), sizeof(PointT), in(original),
        original.size(), inlier_threshold_, max_iterations_, out(tmp), &n_out, best_transformation_.data());
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  double inlier_threshold_;
  int max_iterations_;
  PointCloudConstPtr input_, target_;
  Matrix4f best_transformation_;
};

// correspondence_rejection_surface_normal.h / impl/_surface_normal.hpp:44-66.  The reference keeps the clouds and
// normals in a type-erased DataContainer (initializeDataContainer<PointT, NormalT>() + templated setters); the same
// calls are kept here, the normals are what the rejection reads.
class CorrespondenceRejectorSurfaceNormal : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorSurfaceNormal() : threshold_(1.0) { rejection_name_ = "CorrespondenceRejectorSurfaceNormal"; }
  void setThreshold(double threshold) { threshold_ = threshold; }
  double getThreshold() const { return threshold_; }
  template <typename PointT, typename NormalT>
  void initializeDataContainer() { initialized_ = true; }
  template <typename PointT>
  void setInputCloud(const typename pcl::PointCloud<PointT>::ConstPtr &) { check("setInputCloud"); }
  template <typename PointT>
  void setInputTarget(const typename pcl::PointCloud<PointT>::ConstPtr &) { check("setInputTarget"); }
  template <typename PointT, typename NormalT>
  void setInputNormals(const typename pcl::PointCloud<NormalT>::ConstPtr &normals) {
    if (check("setInputNormals")) pack(*normals, src_normals_);
  }
  template <typename PointT, typename NormalT>
  void setTargetNormals(const typename pcl::PointCloud<NormalT>::ConstPtr &normals) {
    if (check("setTargetNormals")) pack(*normals, tgt_normals_);
  }
  void getRemainingCorrespondences(const pcl::Correspondences &original, pcl::Correspondences &remaining) override {
    if (!initialized_) {  // _surface_normal.hpp:49-53
      PCL_ERROR("DataContainer object is not initialized!\n");
      return;
    }
    if (src_normals_.empty() || tgt_normals_.empty()) {
      PCL_ERROR("[pcl::registration::%s] Normals are not set for the input and target point clouds\n",
                rejection_name_.c_str());
      return;
    }
    pcl::Correspondences tmp(original.size());
    size_t n_out = 0;
    const int rc = pclmi_reject_surface_normal(in(original), original.size(), src_normals_.data(),
                                               src_normals_.size() / 3, 12, tgt_normals_.data(), tgt_normals_.size() / 3,
                                               12, threshold_, out(tmp), &n_out);
    finish(rc, n_out, tmp);
    remaining.swap(tmp);
  }

 protected:
  bool check(const char *fn) const {
    if (!initialized_)
      PCL_ERROR("[pcl::registration::%s::%s] Initilize the data container object by calling intializeDataContainer () "
                "before using this function.\n", rejection_name_.c_str(), fn);
    return initialized_;
  }
  template <typename CloudT>
  static void pack(const CloudT &cloud, std::vector<float> &out) {
    out.resize(cloud.points.size() * 3);
    for (size_t i = 0; i < cloud.points.size(); ++i) {
      out[3 * i + 0] = cloud.points[i].normal_x;
      out[3 * i + 1] = cloud.points[i].normal_y;
      out[3 * i + 2] = cloud.points[i].normal_z;
    }
  }
  double threshold_;
  bool initialized_ = false;
  std::vector<float> src_normals_, tgt_normals_;
};

}  // namespace registration
}  // namespace pcl
#endif
