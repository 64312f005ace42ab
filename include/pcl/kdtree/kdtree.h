// pcl/kdtree/kdtree.h - abstract pcl::KdTree<PointT> (kdtree/include/pcl/kdtree/kdtree.h:55-372).
#ifndef PCLMI_FACADE_KDTREE_KDTREE_H_
#define PCLMI_FACADE_KDTREE_KDTREE_H_

#include <cassert>
#include <climits>
#include <string>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_representation.h>

namespace pcl {

template <typename PointT>
class KdTree {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef std::shared_ptr<PointCloud> PointCloudPtr;
  typedef std::shared_ptr<const PointCloud> PointCloudConstPtr;
  typedef pcl::PointRepresentation<PointT> PointRepresentation;
  typedef std::shared_ptr<const PointRepresentation> PointRepresentationConstPtr;
  typedef std::shared_ptr<KdTree<PointT> > Ptr;
  typedef std::shared_ptr<const KdTree<PointT> > ConstPtr;

  explicit KdTree(bool sorted = true)
      : epsilon_(0.0f), min_pts_(1), sorted_(sorted), point_representation_(new DefaultPointRepresentation<PointT>) {}
  virtual ~KdTree() {}

  virtual void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr &indices = IndicesConstPtr()) {
    input_ = cloud;
    indices_ = indices;
  }
  IndicesConstPtr getIndices() const { return indices_; }
  PointCloudConstPtr getInputCloud() const { return input_; }

  void setPointRepresentation(const PointRepresentationConstPtr &point_representation) {  // :111-117
    point_representation_ = point_representation;
    setInputCloud(input_, indices_);  // rebuilds the tree in derived classes
  }
  PointRepresentationConstPtr getPointRepresentation() const { return point_representation_; }

  virtual int nearestKSearch(const PointT &p_q, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const = 0;

  virtual int nearestKSearch(const PointCloud &cloud, int index, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const {  // :157-163
    assert(index >= 0 && index < static_cast<int>(cloud.points.size()) && "Out-of-bounds error in nearestKSearch!");
    return nearestKSearch(cloud.points[index], k, k_indices, k_sqr_distances);
  }

  template <typename PointTDiff>
  int nearestKSearchT(const PointTDiff &point, int k, std::vector<int> &k_indices,
                      std::vector<float> &k_sqr_distances) const {  // :174-183
    PointT p;
    p.x = point.x;
    p.y = point.y;
    p.z = point.z;
    return nearestKSearch(p, k, k_indices, k_sqr_distances);
  }

  virtual int nearestKSearch(int index, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const {  // :204-218
    if (!indices_) {
      assert(index >= 0 && index < static_cast<int>(input_->points.size()));
      return nearestKSearch(input_->points[index], k, k_indices, k_sqr_distances);
    }
    assert(index >= 0 && index < static_cast<int>(indices_->size()));
    return nearestKSearch(input_->points[(*indices_)[index]], k, k_indices, k_sqr_distances);
  }

  virtual int radiusSearch(const PointT &p_q, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const = 0;

  virtual int radiusSearch(const PointCloud &cloud, int index, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {  // :251-257
    assert(index >= 0 && index < static_cast<int>(cloud.points.size()) && "Out-of-bounds error in radiusSearch!");
    return radiusSearch(cloud.points[index], radius, k_indices, k_sqr_distances, max_nn);
  }

  template <typename PointTDiff>
  int radiusSearchT(const PointTDiff &point, double radius, std::vector<int> &k_indices,
                    std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {  // :270-279
    PointT p;
    p.x = point.x;
    p.y = point.y;
    p.z = point.z;
    return radiusSearch(p, radius, k_indices, k_sqr_distances, max_nn);
  }

  virtual int radiusSearch(int index, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {  // :302-316
    if (!indices_) {
      assert(index >= 0 && index < static_cast<int>(input_->points.size()));
      return radiusSearch(input_->points[index], radius, k_indices, k_sqr_distances, max_nn);
    }
    assert(index >= 0 && index < static_cast<int>(indices_->size()));
    return radiusSearch(input_->points[(*indices_)[index]], radius, k_indices, k_sqr_distances, max_nn);
  }

  virtual void setEpsilon(float eps) { epsilon_ = eps; }
  float getEpsilon() const { return epsilon_; }
  void setMinPts(int min_pts) { min_pts_ = min_pts; }
  int getMinPts() const { return min_pts_; }

 protected:
  PointCloudConstPtr input_;
  IndicesConstPtr indices_;
  float epsilon_;
  int min_pts_;
  bool sorted_;
  PointRepresentationConstPtr point_representation_;
  virtual std::string getName() const = 0;
};

}  // namespace pcl
#endif
