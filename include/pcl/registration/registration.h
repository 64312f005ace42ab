// pcl/registration/registration.h - pcl::Registration<PointSource, PointTarget>
// (registration/include/pcl/registration/registration.h:58-382, impl/registration.hpp:41-168).
#ifndef PCLMI_FACADE_REGISTRATION_REGISTRATION_H_
#define PCLMI_FACADE_REGISTRATION_REGISTRATION_H_

#include <cmath>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include <pcl/common/matrix4.h>
#include <pcl/console/print.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/pcl_base.h>
#include <pcl/registration/transformation_estimation.h>

namespace pcl {

template <typename PointSource, typename PointTarget>
class Registration : public PCLBase<PointSource> {
 public:
  using PCLBase<PointSource>::initCompute;
  using PCLBase<PointSource>::deinitCompute;
  using PCLBase<PointSource>::input_;
  using PCLBase<PointSource>::indices_;
  using PCLBase<PointSource>::fake_indices_;

  typedef std::shared_ptr<Registration<PointSource, PointTarget> > Ptr;
  typedef pcl::KdTreeFLANN<PointTarget> KdTree;
  typedef std::shared_ptr<KdTree> KdTreePtr;
  typedef pcl::PointCloud<PointSource> PointCloudSource;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef pcl::PointCloud<PointTarget> PointCloudTarget;
  typedef typename PointCloudTarget::Ptr PointCloudTargetPtr;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef typename KdTree::PointRepresentationConstPtr PointRepresentationConstPtr;
  typedef pcl::registration::TransformationEstimation<PointSource, PointTarget> TransformationEstimation;
  typedef typename TransformationEstimation::Ptr TransformationEstimationPtr;

  // defaults: registration.h:88-108
  Registration()
      : reg_name_(), tree_(new KdTree), nr_iterations_(0), max_iterations_(10), ransac_iterations_(0), target_(),
        transformation_epsilon_(0.0), euclidean_fitness_epsilon_(-std::numeric_limits<double>::max()),
        corr_dist_threshold_(std::sqrt(std::numeric_limits<double>::max())), inlier_threshold_(0.05),
        converged_(false), min_number_correspondences_(3) {}
  virtual ~Registration() {}

  void setTransformationEstimation(const TransformationEstimationPtr &te) { transformation_estimation_ = te; }

  // impl/registration.hpp:41-57: deep copy with data[3] = 1 and tree build - both inside pclmi_tree_create
  virtual void setInputTarget(const PointCloudTargetConstPtr &cloud) {
    if (!cloud || cloud->points.empty()) {
      PCL_ERROR("[pcl::%s::setInputTarget] Invalid or empty point cloud dataset given!\n", getClassName().c_str());
      return;
    }
    target_ = cloud;
    tree_->setInputCloud(target_);
  }
  PointCloudTargetConstPtr const getInputTarget() { return target_; }

  Matrix4f getFinalTransformation() { return final_transformation_; }
  Matrix4f getLastIncrementalTransformation() { return transformation_; }
  void setMaximumIterations(int nr_iterations) { max_iterations_ = nr_iterations; }
  int getMaximumIterations() { return max_iterations_; }
  void setRANSACIterations(int ransac_iterations) { ransac_iterations_ = ransac_iterations; }
  double getRANSACIterations() { return ransac_iterations_; }
  void setRANSACOutlierRejectionThreshold(double inlier_threshold) { inlier_threshold_ = inlier_threshold; }
  double getRANSACOutlierRejectionThreshold() { return inlier_threshold_; }
  void setMaxCorrespondenceDistance(double distance_threshold) { corr_dist_threshold_ = distance_threshold; }
  double getMaxCorrespondenceDistance() { return corr_dist_threshold_; }
  void setTransformationEpsilon(double epsilon) { transformation_epsilon_ = epsilon; }
  double getTransformationEpsilon() { return transformation_epsilon_; }
  void setEuclideanFitnessEpsilon(double epsilon) { euclidean_fitness_epsilon_ = epsilon; }
  double getEuclideanFitnessEpsilon() { return euclidean_fitness_epsilon_; }
  void setPointRepresentation(const PointRepresentationConstPtr &point_representation) {
    point_representation_ = point_representation;
    tree_->setPointRepresentation(point_representation_);
  }

  /** \brief registration.h registerVisualizationCallback: the function is called after every iteration with the
    * moved source, the source indices of the pairs the estimator was given, the target and the matching target
    * indices (icp.hpp:182-183).  With a callback registered the device loop runs one iteration per launch and the
    * cloud travels to the host after each (pclmi_icp_align_observed). */
  typedef std::function<void(const pcl::PointCloud<PointSource> &, const std::vector<int> &,
                             const pcl::PointCloud<PointTarget> &, const std::vector<int> &)> UpdateVisualizerFn;
  bool registerVisualizationCallback(UpdateVisualizerFn &visualizerCallback) {
    if (visualizerCallback) {
      update_visualizer_ = visualizerCallback;
      return true;
    }
    return false;
  }

  // impl/registration.hpp:71-109
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double score = std::numeric_limits<double>::max();
    if (!input_ || !tree_->getDeviceTree()) return score;
    if (point_representation_ && !point_representation_->isTrivial()) {
      // the pairs are found in the representation's space (registration.hpp:91 through the tree), the score is
      // the squared distance between the POINTS of each pair (:87-101): move the input on the host
      // (transforms.hpp:184-191), vectorise it, ONE batched 1-NN call, fp64 sum over the pairs within max_range
      const Matrix4f &T = final_transformation_;
      const size_t n = input_->points.size();
      std::vector<float> q(n * 3), raw(n * 3);
      for (size_t i = 0; i < n; ++i) {
        PointTarget moved = PointTarget();
        const PointSource &p = input_->points[i];
        moved.x = raw[3 * i] = ((T(0, 0) * p.x + T(0, 1) * p.y) + T(0, 2) * p.z) + T(0, 3);
        moved.y = raw[3 * i + 1] = ((T(1, 0) * p.x + T(1, 1) * p.y) + T(1, 2) * p.z) + T(1, 3);
        moved.z = raw[3 * i + 2] = ((T(2, 0) * p.x + T(2, 1) * p.y) + T(2, 2) * p.z) + T(2, 3);
        tree_->toXYZ(moved, &q[3 * i]);
      }
      std::vector<int> idx;
      std::vector<float> d2;
      if (tree_->nearestKSearchBatch(q.data(), n, 12, 1, idx, d2) != 1) return score;
      double sum = 0.0;
      int nr = 0;
      for (size_t i = 0; i < n; ++i) {
        if (idx[i] < 0 || d2[i] > max_range) continue;  // registration.hpp:94
        const PointTarget &t = target_->points[idx[i]];
        const float dx = raw[3 * i] - t.x, dy = raw[3 * i + 1] - t.y, dz = raw[3 * i + 2] - t.z;
        sum += std::fabs(static_cast<double>((dx * dx + dy * dy) + dz * dz));
        ++nr;
      }
      return nr > 0 ? sum / nr : score;
    }
    const int rc = pclmi_fitness_score(tree_->getDeviceTree(), reinterpret_cast<const float *>(input_->points.data()),
                                       input_->points.size(), sizeof(PointSource), final_transformation_.data(),
                                       max_range, &score);
    if (rc != PCLMI_OK) PCL_ERROR("[pcl::%s::getFitnessScore] %s\n", getClassName().c_str(), pclmi_last_error());
    return score;
  }
  // impl/registration.hpp:60-68
  double getFitnessScore(const std::vector<float> &distances_a, const std::vector<float> &distances_b) {
    const size_t nr_elem = std::min(distances_a.size(), distances_b.size());
    float s = 0.f;
    for (size_t i = 0; i < nr_elem; ++i) s += distances_a[i] - distances_b[i];
    return static_cast<double>(s) / static_cast<double>(nr_elem);
  }

  bool hasConverged() { return converged_; }

  void align(PointCloudSource &output) { align(output, Matrix4f::Identity()); }

  // impl/registration.hpp:119-168
  void align(PointCloudSource &output, const Matrix4f &guess) {
    if (!initCompute()) return;
    if (!target_) {
      PCL_WARN("[pcl::%s::compute] No input target dataset was given!\n", getClassName().c_str());
      return;
    }
    // output <- input[indices] (registration.hpp:133-150)
    if (indices_->size() != output.points.size()) output.points.resize(indices_->size());
    output.header = input_->header;
    if (indices_->size() == input_->points.size()) {
      output.width = input_->width;
      output.height = input_->height;
      output.is_dense = input_->is_dense;
    } else {
      output.width = static_cast<uint32_t>(indices_->size());
      output.height = 1;
      output.is_dense = false;
    }
    for (size_t i = 0; i < indices_->size(); ++i) output.points[i] = input_->points[(*indices_)[i]];
    converged_ = false;
    final_transformation_.setIdentity();
    transformation_.setIdentity();
    previous_transformation_.setIdentity();
    computeTransformation(output, guess);
    deinitCompute();
  }

  const std::string &getClassName() const { return reg_name_; }

 protected:
  virtual void computeTransformation(PointCloudSource &output, const Matrix4f &guess) = 0;

  std::string reg_name_;
  KdTreePtr tree_;
  int nr_iterations_;
  int max_iterations_;
  int ransac_iterations_;
  PointCloudTargetConstPtr target_;
  Matrix4f final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_;
  double euclidean_fitness_epsilon_;
  double corr_dist_threshold_;
  double inlier_threshold_;
  bool converged_;
  int min_number_correspondences_;
  TransformationEstimationPtr transformation_estimation_;
  PointRepresentationConstPtr point_representation_;
  UpdateVisualizerFn update_visualizer_;
};

}  // namespace pcl
#endif
