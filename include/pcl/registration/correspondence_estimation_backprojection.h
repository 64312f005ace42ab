// pcl/registration/correspondence_estimation_backprojection.h - CorrespondenceEstimationBackProjection
// (registration/include/pcl/registration/correspondence_estimation_backprojection.h:54-160,
//  impl/correspondence_estimation_backprojection.hpp:44-107): for every source point, among its k nearest target
// points the arg-min of the reference's normal-compatibility weight.  One batched device call
// (pclmi_correspondences_back_projection) instead of a per-point tree query; the reference's carried-over
// min_index (:71) is reproduced on the device by a scan over function composition.
#ifndef PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_BACKPROJECTION_H_
#define PCLMI_FACADE_REGISTRATION_CORRESPONDENCE_ESTIMATION_BACKPROJECTION_H_

#include <pcl/registration/correspondence_estimation.h>

namespace pcl {
namespace registration {

template <typename PointSource, typename PointTarget, typename NormalT>
class CorrespondenceEstimationBackProjection : public CorrespondenceEstimation<PointSource, PointTarget> {
 public:
  using CorrespondenceEstimation<PointSource, PointTarget>::input_;
  using CorrespondenceEstimation<PointSource, PointTarget>::indices_;
  using CorrespondenceEstimation<PointSource, PointTarget>::fake_indices_;
  using CorrespondenceEstimation<PointSource, PointTarget>::tree_;
  using CorrespondenceEstimation<PointSource, PointTarget>::target_;
  using CorrespondenceEstimation<PointSource, PointTarget>::corr_name_;
  typedef typename pcl::PointCloud<NormalT>::Ptr NormalsPtr;
  typedef std::shared_ptr<CorrespondenceEstimationBackProjection<PointSource, PointTarget, NormalT> > Ptr;

  /** \brief k defaults to 10; the reference names the class "...NormalShooting" here too (:96-102) */
  CorrespondenceEstimationBackProjection() : k_(10) { corr_name_ = "CorrespondenceEstimationNormalShooting"; }

  inline void setSourceNormals(const NormalsPtr &normals) { source_normals_ = normals; }
  inline NormalsPtr getSourceNormals() const { return source_normals_; }
  inline void setTargetNormals(const NormalsPtr &normals) { target_normals_ = normals; }
  inline NormalsPtr getTargetNormals() const { return target_normals_; }
  inline void setKSearch(unsigned int k) { k_ = k; }
  inline unsigned int getKSearch() const { return k_; }

  void determineCorrespondences(pcl::Correspondences &correspondences,
                                double max_distance = std::numeric_limits<double>::max()) override {
    correspondences.clear();
    if (!source_normals_ || !target_normals_) {  // impl/...backprojection.hpp:46-50
      PCL_WARN("[pcl::%s::compute] Datasets containing normals for source/target have not been given!\n",
               corr_name_.c_str());
      return;
    }
    if (!input_ || source_normals_->points.size() != input_->points.size()) {
      PCL_ERROR("[pcl::%s::compute] The source normals do not match the source cloud!\n", corr_name_.c_str());
      return;
    }
    if (!this->initComputeTarget()) return;
    if (target_normals_->points.size() != target_->points.size()) {
      PCL_ERROR("[pcl::%s::compute] The target normals do not match the target cloud!\n", corr_name_.c_str());
      this->deinitCompute();
      return;
    }
    const size_t n = fake_indices_ ? input_->points.size() : indices_->size();
    correspondences.resize(n);
    size_t n_out = 0;
    const int rc = pclmi_correspondences_back_projection(
        tree_->getDeviceTree(), &input_->points[0].x, input_->points.size(), sizeof(PointSource),
        &source_normals_->points[0].normal_x, sizeof(NormalT), &target_normals_->points[0].normal_x,
        target_normals_->points.size(), sizeof(NormalT),
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
  NormalsPtr source_normals_;
  NormalsPtr target_normals_;
  unsigned int k_;
};

}  // namespace registration
}  // namespace pcl
#endif
