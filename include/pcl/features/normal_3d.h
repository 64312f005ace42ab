// pcl/features/normal_3d.h - pcl::NormalEstimation and the free plane-fit helpers
// (features/include/pcl/features/normal_3d.h:58-330, impl/normal_3d.hpp:46-98) over the MI355X library:
// compute() is ONE pclmi_normals call for the whole index list (batched search + fused plane fit on the
// device); computePointNormal on a single list is pclmi_point_normal (host).
#ifndef PCLMI_FACADE_FEATURES_NORMAL_3D_H_
#define PCLMI_FACADE_FEATURES_NORMAL_3D_H_

#include <cmath>
#include <limits>
#include <vector>

#include <pcl/common/matrix4.h>
#include <pcl/features/feature.h>
#include <pcl/point_types.h>
#include <pclmi.h>

namespace pcl {

/** \brief normal_3d.h:58-75 - plane of the whole cloud: (a, b, c, d) with ax + by + cz + d = 0, and curvature */
template <typename PointT>
inline void computePointNormal(const pcl::PointCloud<PointT> &cloud, Vector4f &plane_parameters, float &curvature) {
  float pl[4];
  if (cloud.points.empty() ||
      pclmi_point_normal(reinterpret_cast<const float *>(&cloud.points[0]), cloud.points.size(), sizeof(PointT), nullptr, 0,
                         pl, &curvature) != PCLMI_OK) {
    pl[0] = pl[1] = pl[2] = pl[3] = curvature = std::numeric_limits<float>::quiet_NaN();
  }
  for (int d = 0; d < 4; ++d) plane_parameters[d] = pl[d];
}

/** \brief normal_3d.h:88-104 - plane of an index list */
template <typename PointT>
inline void computePointNormal(const pcl::PointCloud<PointT> &cloud, const std::vector<int> &indices,
                               Vector4f &plane_parameters, float &curvature) {
  float pl[4];
  if (cloud.points.empty() || indices.empty() ||
      pclmi_point_normal(reinterpret_cast<const float *>(&cloud.points[0]), cloud.points.size(), sizeof(PointT),
                         &indices[0], indices.size(), pl, &curvature) != PCLMI_OK) {
    pl[0] = pl[1] = pl[2] = pl[3] = curvature = std::numeric_limits<float>::quiet_NaN();
  }
  for (int d = 0; d < 4; ++d) plane_parameters[d] = pl[d];
}

/** \brief normal_3d.h:114-131 - flips the plane normal towards the viewpoint and recomputes d through `point`.
 *  (The reference builds the view vector as (vp_x - x, vp_y - x, vp_z - z), with point.x in the second component;
 *  the documented intent, (vp - point), is implemented - as its own three-float overload does.) */
template <typename PointT>
inline void flipNormalTowardsViewpoint(const PointT &point, float vp_x, float vp_y, float vp_z, Vector4f &normal) {
  const float cos_theta = (vp_x - point.x) * normal[0] + (vp_y - point.y) * normal[1] + (vp_z - point.z) * normal[2];
  if (cos_theta < 0) {
    normal[0] *= -1;
    normal[1] *= -1;
    normal[2] *= -1;
    normal[3] = -1 * (normal[0] * point.x + normal[1] * point.y + normal[2] * point.z);
  }
}

/** \brief normal_3d.h:162-181 */
template <typename PointT>
inline void flipNormalTowardsViewpoint(const PointT &point, float vp_x, float vp_y, float vp_z, float &nx, float &ny,
                                       float &nz) {
  vp_x -= point.x;
  vp_y -= point.y;
  vp_z -= point.z;
  const float cos_theta = (vp_x * nx + vp_y * ny + vp_z * nz);
  if (cos_theta < 0) {
    nx *= -1;
    ny *= -1;
    nz *= -1;
  }
}

template <typename PointInT, typename PointOutT>
class NormalEstimation : public Feature<PointInT, PointOutT> {
 public:
  typedef std::shared_ptr<NormalEstimation<PointInT, PointOutT> > Ptr;
  using Feature<PointInT, PointOutT>::feature_name_;
  using Feature<PointInT, PointOutT>::indices_;
  using Feature<PointInT, PointOutT>::input_;
  using Feature<PointInT, PointOutT>::surface_;
  using Feature<PointInT, PointOutT>::k_;
  using Feature<PointInT, PointOutT>::search_radius_;
  typedef typename Feature<PointInT, PointOutT>::PointCloudOut PointCloudOut;
  typedef typename Feature<PointInT, PointOutT>::PointCloudInConstPtr PointCloudConstPtr;

  NormalEstimation() : vpx_(0), vpy_(0), vpz_(0), use_sensor_origin_(true) { feature_name_ = "NormalEstimation"; }
  virtual ~NormalEstimation() {}

  /** \brief normal_3d.h:231-242 */
  inline void computePointNormal(const pcl::PointCloud<PointInT> &cloud, const std::vector<int> &indices,
                                 Vector4f &plane_parameters, float &curvature) {
    pcl::computePointNormal(cloud, indices, plane_parameters, curvature);
  }
  /** \brief normal_3d.h:256-269 */
  inline void computePointNormal(const pcl::PointCloud<PointInT> &cloud, const std::vector<int> &indices, float &nx,
                                 float &ny, float &nz, float &curvature) {
    Vector4f pl;
    pcl::computePointNormal(cloud, indices, pl, curvature);
    nx = pl[0];
    ny = pl[1];
    nz = pl[2];
  }

  virtual void setInputCloud(const PointCloudConstPtr &cloud) override {
    input_ = cloud;
    if (use_sensor_origin_) vpx_ = vpy_ = vpz_ = 0.f;  // the facade's clouds carry no sensor pose: the origin
  }
  inline void setViewPoint(float vpx, float vpy, float vpz) {
    vpx_ = vpx;
    vpy_ = vpy;
    vpz_ = vpz;
    use_sensor_origin_ = false;
  }
  inline void getViewPoint(float &vpx, float &vpy, float &vpz) {
    vpx = vpx_;
    vpy = vpy_;
    vpz = vpz_;
  }
  inline void useSensorOriginAsViewPoint() {
    use_sensor_origin_ = true;
    vpx_ = vpy_ = vpz_ = 0.f;
  }

 protected:
  /** \brief impl/normal_3d.hpp:46-98 in one device call */
  void computeFeature(PointCloudOut &output) override {
    const size_t nq = indices_->size();
    std::vector<float> q(nq * 3), res(nq * 4, std::numeric_limits<float>::quiet_NaN());
    for (size_t i = 0; i < nq; ++i) {
      const PointInT &p = input_->points[(*indices_)[i]];
      q[3 * i + 0] = p.x;
      q[3 * i + 1] = p.y;
      q[3 * i + 2] = p.z;
    }
    const float vp[3] = {vpx_, vpy_, vpz_};
    const pclmi_tree *t = this->surfaceDeviceTree();
    if (!t || pclmi_normals(t, q.data(), nq, 12, k_, search_radius_, vp, res.data()) != PCLMI_OK)
      PCL_ERROR("[pcl::%s::computeFeature] %s\n", feature_name_.c_str(), pclmi_last_error());
    output.is_dense = true;
    for (size_t i = 0; i < nq; ++i) {
      output.points[i].normal_x = res[4 * i + 0];
      output.points[i].normal_y = res[4 * i + 1];
      output.points[i].normal_z = res[4 * i + 2];
      output.points[i].curvature = res[4 * i + 3];
      if (std::isnan(res[4 * i + 0])) output.is_dense = false;
    }
  }

  float vpx_, vpy_, vpz_;
  bool use_sensor_origin_;
};

/** \brief features/include/pcl/features/normal_3d_omp.h: the OpenMP variant - the device batch is already parallel */
template <typename PointInT, typename PointOutT>
class NormalEstimationOMP : public NormalEstimation<PointInT, PointOutT> {
 public:
  explicit NormalEstimationOMP(unsigned int nr_threads = 0) : threads_(nr_threads) { this->feature_name_ = "NormalEstimationOMP"; }
  inline void setNumberOfThreads(unsigned int nr_threads = 0) { threads_ = nr_threads; }

 protected:
  unsigned int threads_;
};

}  // namespace pcl
#endif
