// pcl/registration/icp.h - pcl::IterativeClosestPoint<PointSource, PointTarget>
// (registration/include/pcl/registration/icp.h:87-136, impl/icp.hpp:44-211).
// The whole correspondence loop - 1-NN, threshold, optional RANSAC rejection, centroid /
// cross-covariance reduction, SVD, transform, convergence tests - runs in ONE library call
// (pclmi_icp_align); the source cloud crosses PCIe once in, once out.
// setTransformationEstimation (registration.h:140-160) selects the estimator of icp.hpp:171:
// TransformationEstimationSVD (default) or TransformationEstimationPointToPlaneLLS - then the loop is
// pclmi_icp_align_point_to_plane and the point types must carry normal_x/y/z (e.g. pcl::PointNormal).
#ifndef PCLMI_FACADE_REGISTRATION_ICP_H_
#define PCLMI_FACADE_REGISTRATION_ICP_H_

#include <cmath>
#include <limits>

#include <pcl/registration/registration.h>
#include <pcl/registration/transformation_estimation_svd.h>

namespace pcl {

namespace detail {
// &p.normal_x when the point type has normals, nullptr otherwise
template <typename P>
auto normal_ptr_of(const P *p, int) -> decltype(&p->normal_x) { return &p->normal_x; }
template <typename P>
const float *normal_ptr_of(const P *, long) { return nullptr; }
}  // namespace detail

template <typename PointSource, typename PointTarget>
class IterativeClosestPoint : public Registration<PointSource, PointTarget> {
  typedef Registration<PointSource, PointTarget> Base;

 public:
  typedef typename Base::PointCloudSource PointCloudSource;
  using Base::reg_name_;
  using Base::getClassName;
  using Base::tree_;
  using Base::nr_iterations_;
  using Base::max_iterations_;
  using Base::ransac_iterations_;
  using Base::final_transformation_;
  using Base::transformation_;
  using Base::previous_transformation_;
  using Base::transformation_epsilon_;
  using Base::euclidean_fitness_epsilon_;
  using Base::corr_dist_threshold_;
  using Base::inlier_threshold_;
  using Base::converged_;
  using Base::min_number_correspondences_;
  using Base::transformation_estimation_;

  IterativeClosestPoint() {  // icp.h:99-106
    reg_name_ = "IterativeClosestPoint";
    ransac_iterations_ = 1000;
    transformation_estimation_.reset(new pcl::registration::TransformationEstimationSVD<PointSource, PointTarget>);
  }

  /** \brief Not in the reference: run align() over the ranks of a pclmi_comm (one process per GPU, every process
    * calls align() with the same clouds and parameters): the source is sharded over the ranks, every iteration
    * exchanges the moment sums inside the step kernel (pclmi_icp_align_sharded); with the RANSAC rejection step on
    * (the default) every rank replays the same sampling stream and the inlier counts / refit sums are exchanged as
    * well.  Every rank ends with the same final transformation; rows of the output cloud that
    * belong to other ranks are the input moved by the final transformation in one step.  NULL detaches. */
  void setComm(pclmi_comm *comm) { comm_ = comm; }

  // statistics of the last align()
  int getNumberOfIterations() const { return nr_iterations_; }
  const pclmi_icp_stats &getLastStats() const { return stats_; }

 protected:
  void computeTransformation(PointCloudSource &output, const Matrix4f &guess) override {
    nr_iterations_ = 0;
    std::memset(&stats_, 0, sizeof(stats_));
    const int loop = transformation_estimation_ ? transformation_estimation_->deviceLoop() : -1;
    if (loop < 0) {
      PCL_ERROR("[pcl::%s::computeTransformation] the device loop runs TransformationEstimationSVD or "
                "TransformationEstimationPointToPlaneLLS; this estimator has no device loop\n", getClassName().c_str());
      return;
    }
    if (!tree_->getDeviceTree() || output.points.empty()) return;
    const float *src_normals = detail::normal_ptr_of(output.points.data(), 0);
    const float *tgt_normals = Base::target_ ? detail::normal_ptr_of(Base::target_->points.data(), 0) : nullptr;
    if (loop == 1 && !tgt_normals) {
      PCL_ERROR("[pcl::%s::computeTransformation] TransformationEstimationPointToPlaneLLS needs a target point type "
                "with normals (normal_x, normal_y, normal_z)\n", getClassName().c_str());
      return;
    }
    // Registration::setPointRepresentation: the reference vectorises the moving source point through the
    // representation for every search (registration.h:362-370 -> kdtree.h:204-218 -> impl/kdtree_flann.hpp:91-92) and
    // hands the points themselves to the estimator.  A per-dimension rescale of x, y, z runs inside the device loop
    // (pclmi_icp_align_scaled); anything else cannot be applied to a point that only exists on the device.
    float rep_scale[3] = {1.f, 1.f, 1.f};
    bool scaled = false;
    if (Base::point_representation_ && !Base::point_representation_->isTrivial()) {
      scaled = Base::point_representation_->diagonalScale(rep_scale);
      if (!scaled)
        PCL_WARN("[pcl::%s::computeTransformation] this point representation is not a rescale of x, y, z: it applies "
                 "to the target tree only, the correspondence loop reads the source's raw x, y, z\n",
                 getClassName().c_str());
      else if (loop == 1 || Base::update_visualizer_) {
        PCL_ERROR("[pcl::%s::computeTransformation] a rescaling point representation is supported with "
                  "TransformationEstimationSVD and without a visualization callback\n", getClassName().c_str());
        return;
      }
    }
    pclmi_icp_params p;
    pclmi_icp_default_params(&p);
    p.max_iterations = max_iterations_;
    p.ransac_iterations = ransac_iterations_;
    p.transformation_epsilon = transformation_epsilon_;
    p.euclidean_fitness_epsilon = euclidean_fitness_epsilon_;
    p.max_correspondence_distance = corr_dist_threshold_;
    p.ransac_outlier_threshold = inlier_threshold_;
    p.min_correspondences = min_number_correspondences_;
    std::vector<float> aligned(output.points.size() * 4);
    // `output` already holds input[indices] (Registration::align); the library sets w = 1, applies the
    // guess (icp.hpp:60-66) and iterates
    const bool sharded = comm_ && pclmi_comm_size(comm_) > 1;
    if (sharded && Base::update_visualizer_) {
      PCL_ERROR("[pcl::%s::computeTransformation] a visualization callback sees the whole source every iteration: it "
                "cannot be combined with setComm\n", getClassName().c_str());
      return;
    }
    if (sharded)
      for (size_t i = 0; i < aligned.size(); ++i) aligned[i] = std::numeric_limits<float>::quiet_NaN();
    const float *src_xyz = reinterpret_cast<const float *>(output.points.data());
    ObserverState os{this, &output, {}, {}};
    const int rc =
        Base::update_visualizer_
            ? pclmi_icp_align_observed(&p, tree_->getDeviceTree(), src_xyz, output.points.size(), sizeof(PointSource),
                                       loop == 1 ? src_normals : nullptr, sizeof(PointSource),
                                       loop == 1 ? tgt_normals : nullptr, Base::target_->points.size(),
                                       sizeof(PointTarget), guess.isIdentity() ? nullptr : guess.data(), &observe, &os,
                                       final_transformation_.data(), aligned.data(), &stats_)
        : loop == 1 ? pclmi_icp_align_point_to_plane(&p, tree_->getDeviceTree(), src_xyz, output.points.size(),
                                                   sizeof(PointSource), src_normals, sizeof(PointSource), tgt_normals,
                                                   Base::target_->points.size(), sizeof(PointTarget),
                                                   guess.isIdentity() ? nullptr : guess.data(), comm_,
                                                   final_transformation_.data(), aligned.data(), &stats_)
        : scaled ? pclmi_icp_align_scaled(&p, tree_->getDeviceTree(), rep_scale,
                                          reinterpret_cast<const float *>(Base::target_->points.data()),
                                          Base::target_->points.size(), sizeof(PointTarget), src_xyz,
                                          output.points.size(), sizeof(PointSource),
                                          guess.isIdentity() ? nullptr : guess.data(), comm_,
                                          final_transformation_.data(), aligned.data(), &stats_)
        : sharded ? pclmi_icp_align_sharded(&p, tree_->getDeviceTree(), src_xyz, output.points.size(), sizeof(PointSource),
                                          guess.isIdentity() ? nullptr : guess.data(), comm_,
                                          final_transformation_.data(), aligned.data(), &stats_)
                : pclmi_icp_align(&p, tree_->getDeviceTree(), src_xyz, output.points.size(), sizeof(PointSource),
                                  guess.isIdentity() ? nullptr : guess.data(), final_transformation_.data(),
                                  aligned.data(), &stats_);
    nr_iterations_ = stats_.iterations;
    converged_ = stats_.converged != 0;
    // transformation_ = the last iteration's incremental transform (icp.hpp:171), previous_transformation_ the one
    // before it is not kept by the device loop: getLastIncrementalTransformation() is what callers read
    for (int i = 0; i < 16; ++i) transformation_.data()[i] = stats_.last_transform[i];
    if (rc == PCLMI_ERR_NOT_ENOUGH) {  // icp.hpp:103-108
      PCL_ERROR("[pcl::%s::computeTransformation] Not enough correspondences found. Relax your threshold parameters.\n",
                getClassName().c_str());
    } else if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::%s::computeTransformation] %s\n", getClassName().c_str(), pclmi_last_error());
      return;
    }
    for (size_t i = 0; i < output.points.size(); ++i) {
      if (sharded && std::isnan(aligned[i * 4 + 0])) {  // another rank's row: the input moved by the final transform
        const float x = output.points[i].x, y = output.points[i].y, z = output.points[i].z;
        const float *F = final_transformation_.data();
        if (std::isfinite(x) && std::isfinite(y) && std::isfinite(z)) {
          aligned[i * 4 + 0] = ((F[0] * x + F[1] * y) + F[2] * z) + F[3];
          aligned[i * 4 + 1] = ((F[4] * x + F[5] * y) + F[6] * z) + F[7];
          aligned[i * 4 + 2] = ((F[8] * x + F[9] * y) + F[10] * z) + F[11];
        } else {
          aligned[i * 4 + 0] = x;
          aligned[i * 4 + 1] = y;
          aligned[i * 4 + 2] = z;
        }
      }
      output.points[i].x = aligned[i * 4 + 0];
      output.points[i].y = aligned[i * 4 + 1];
      output.points[i].z = aligned[i * 4 + 2];
      output.points[i].data[3] = 1.0f;  // registration.hpp:152-163
    }
    PCL_DEBUG("[pcl::%s::computeTransformation] %d iterations, %d correspondences (%d after rejection), converged %d, "
              "transform delta %g, fitness delta %g\n", getClassName().c_str(), stats_.iterations,
              stats_.last_correspondences, stats_.last_inliers, stats_.converged, stats_.transform_delta,
              stats_.fitness_delta);
  }

  // trampoline of the per-iteration observer: rebuilds the reference's callback arguments
  struct ObserverState {
    IterativeClosestPoint *self;
    PointCloudSource *output;
    std::vector<int> si, ti;
  };
  static void observe(void *user, int, const float *cloud, const int32_t *src_idx, const int32_t *tgt_idx, size_t n) {
    ObserverState *o = static_cast<ObserverState *>(user);
    for (size_t i = 0; i < o->output->points.size(); ++i) {
      o->output->points[i].x = cloud[4 * i + 0];
      o->output->points[i].y = cloud[4 * i + 1];
      o->output->points[i].z = cloud[4 * i + 2];
    }
    o->si.assign(src_idx, src_idx + n);
    o->ti.assign(tgt_idx, tgt_idx + n);
    o->self->update_visualizer_(*o->output, o->si, *o->self->target_, o->ti);
  }

  pclmi_icp_stats stats_;
  pclmi_comm *comm_ = nullptr;
};

}  // namespace pcl
#endif
