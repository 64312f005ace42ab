// pcl/registration/correspondence_estimation_normal_shooting.h - CorrespondenceEstimationNormalShooting
// (registration/include/pcl/registration/correspondence_estimation_normal_shooting.h:60-195,
//  impl/correspondence_estimation_normal_shooting.hpp:60-165): for every source point, among its k nearest
// target points the one closest to the line along the source normal.  One batched device call
// (pclmi_correspondences_normal_shooting) instead of a per-point tree query.
#ifndef PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_NORMAL_SHOOTING_H_
#define PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_NORMAL_SHOOTING_H_

#include <pcl/registration/correspondence_estimation.h>

namespace pcl {
namespace registration {

template <typename PointSource, typename PointTarget, typename NormalT>
class CorrespondenceEstimationNormalShooting : public CorrespondenceEstimation<PointSource, PointTarget> {
 public:
  using CorrespondenceEstimation<PointSource, PointTarget>::input_;
  using CorrespondenceEstimation<PointSource, PointTarget>::indices_;
  using CorrespondenceEstimation<PointSource, PointTarget>::fake_indices_;
  using CorrespondenceEstimation<PointSource, PointTarget>::tree_;
  using CorrespondenceEstimation<PointSource, PointTarget>::corr_name_;
  typedef pcl::PointCloud<NormalT> PointCloudNormals;
  typedef typename PointCloudNormals::Ptr NormalsPtr;
  typedef typename PointCloudNormals::ConstPtr NormalsConstPtr;
  typedef std::shared_ptr<CorrespondenceEstimationNormalShooting<PointSource, PointTarget, NormalT> > Ptr;

  /** \brief k defaults to 10 (correspondence_estimation_normal_shooting.h:105-110) */
  CorrespondenceEstimationNormalShooting() : k_(10) { corr_name_ = "CorrespondenceEstimationNormalShooting"; }

  inline void setSourceNormals(const NormalsConstPtr &normals) { source_normals_ = normals; }
  inline NormalsConstPtr getSourceNormals() const { return source_normals_; }
  inline void setKSearch(unsigned int k) { k_ = k; }
  inline unsigned int getKSearch() const { return k_; }

  void determineCorrespondences(pcl::Correspondences &correspondences,
                                double max_distance = std::numeric_limits<double>::max()) override {
    correspondences.clear();
    if (!source_normals_ || !input_ || source_normals_->points.size() != input_->points.size()) {
      PCL_WARN("[pcl::%s::determineCorrespondences] Datasets containing normals for source have not been given!\n",
               corr_name_.c_str());
      return;
    }
    if (!this->initComputeTarget()) return;
    const size_t n = fake_indices_ ? input_->points.size() : indices_->size();
    correspondences.resize(n);
    size_t n_out = 0;
    const int rc = pclmi_correspondences_normal_shooting(
        tree_->getDeviceTree(), &input_->points[0].x, input_->points.size(), sizeof(PointSource),
        &source_normals_->points[0].normal_x, sizeof(NormalT),
        fake_indices_ ? nullptr : reinterpret_cast<const int32_t *>(indices_->data()), fake_indices_ ? 0 : indices_->size(),
        static_cast<int>(k_), max_distance, reinterpret_cast<pclmi_correspondence *>(correspondences.data()), &n_out);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::%s::determineCorrespondences] %s\n", corr_name_.c_str(), pclmi_last_error());
      n_out = 0;
    }
    correspondences.resize(n_out);
    this->deinitCompute();
  }

 private:
  NormalsConstPtr source_normals_;
  unsigned int k_;
};

}  // namespace registration
}  // namespace pcl
#endif
