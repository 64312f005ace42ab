// pcl/search/search.h - abstract pcl::search::Search<PointT>
// (search/include/pcl/search/search.h:72-500), including the batched overloads.
#ifndef PCLMI_FACADE_SEARCH_SEARCH_H_
#define PCLMI_FACADE_SEARCH_SEARCH_H_

#include <algorithm>
#include <cassert>
#include <string>
#include <vector>

#include <pcl/point_cloud.h>

namespace pcl {
namespace search {

template <typename PointT>
class Search {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef typename PointCloud::ConstPtr PointCloudConstPtr;
  typedef std::shared_ptr<pcl::search::Search<PointT> > Ptr;
  typedef std::shared_ptr<const pcl::search::Search<PointT> > ConstPtr;
  typedef pcl::IndicesPtr IndicesPtr;
  typedef pcl::IndicesConstPtr IndicesConstPtr;

  explicit Search(const std::string &name = "", bool sorted = false) : sorted_results_(sorted), name_(name) {}
  virtual ~Search() {}

  virtual const std::string &getName() const { return name_; }
  virtual void setSortedResults(bool sorted) { sorted_results_ = sorted; }

  virtual void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr &indices = IndicesConstPtr()) {
    input_ = cloud;
    indices_ = indices;
  }
  virtual PointCloudConstPtr getInputCloud() const { return input_; }
  virtual IndicesConstPtr getIndices() const { return indices_; }

  virtual int nearestKSearch(const PointT &point, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const = 0;

  template <typename PointTDiff>
  int nearestKSearchT(const PointTDiff &point, int k, std::vector<int> &k_indices,
                      std::vector<float> &k_sqr_distances) const {  // search.h:272-283
    PointT p;
    p.x = point.x;
    p.y = point.y;
    p.z = point.z;
    return nearestKSearch(p, k, k_indices, k_sqr_distances);
  }

  virtual int nearestKSearch(const PointCloud &cloud, int index, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const {
    assert(index >= 0 && index < static_cast<int>(cloud.points.size()) && "Out-of-bounds error in nearestKSearch!");
    return nearestKSearch(cloud.points[index], k, k_indices, k_sqr_distances);
  }

  virtual int nearestKSearch(int index, int k, std::vector<int> &k_indices,
                             std::vector<float> &k_sqr_distances) const {
    if (!indices_) {
      assert(index >= 0 && index < static_cast<int>(input_->points.size()));
      return nearestKSearch(input_->points[index], k, k_indices, k_sqr_distances);
    }
    assert(index >= 0 && index < static_cast<int>(indices_->size()));
    return nearestKSearch(input_->points[(*indices_)[index]], k, k_indices, k_sqr_distances);
  }

  // batched form (search.h:244-262): the base class loops; search::KdTree answers it in one call
  virtual void nearestKSearch(const PointCloud &cloud, const std::vector<int> &indices, int k,
                              std::vector<std::vector<int> > &k_indices,
                              std::vector<std::vector<float> > &k_sqr_distances) const {
    if (indices.empty()) {
      k_indices.resize(cloud.size());
      k_sqr_distances.resize(cloud.size());
      for (size_t i = 0; i < cloud.size(); i++)
        nearestKSearch(cloud, static_cast<int>(i), k, k_indices[i], k_sqr_distances[i]);
    } else {
      k_indices.resize(indices.size());
      k_sqr_distances.resize(indices.size());
      for (size_t i = 0; i < indices.size(); i++) nearestKSearch(cloud, indices[i], k, k_indices[i], k_sqr_distances[i]);
    }
  }

  virtual int radiusSearch(const PointT &point, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const = 0;

  template <typename PointTDiff>
  int radiusSearchT(const PointTDiff &point, double radius, std::vector<int> &k_indices,
                    std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {  // search.h:448-459
    PointT p;
    p.x = point.x;
    p.y = point.y;
    p.z = point.z;
    return radiusSearch(p, radius, k_indices, k_sqr_distances, max_nn);
  }

  virtual int radiusSearch(const PointCloud &cloud, int index, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {
    assert(index >= 0 && index < static_cast<int>(cloud.points.size()) && "Out-of-bounds error in radiusSearch!");
    return radiusSearch(cloud.points[index], radius, k_indices, k_sqr_distances, max_nn);
  }

  virtual int radiusSearch(int index, double radius, std::vector<int> &k_indices,
                           std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const {
    if (!indices_) {
      assert(index >= 0 && index < static_cast<int>(input_->points.size()));
      return radiusSearch(input_->points[index], radius, k_indices, k_sqr_distances, max_nn);
    }
    assert(index >= 0 && index < static_cast<int>(indices_->size()));
    return radiusSearch(input_->points[(*indices_)[index]], radius, k_indices, k_sqr_distances, max_nn);
  }

  // batched form (search.h:412-434)
  virtual void radiusSearch(const PointCloud &cloud, const std::vector<int> &indices, double radius,
                            std::vector<std::vector<int> > &k_indices,
                            std::vector<std::vector<float> > &k_sqr_distances, unsigned int max_nn = 0) const {
    if (indices.empty()) {
      k_indices.resize(cloud.size());
      k_sqr_distances.resize(cloud.size());
      for (size_t i = 0; i < cloud.size(); i++)
        radiusSearch(cloud, static_cast<int>(i), radius, k_indices[i], k_sqr_distances[i], max_nn);
    } else {
      k_indices.resize(indices.size());
      k_sqr_distances.resize(indices.size());
      for (size_t i = 0; i < indices.size(); i++)
        radiusSearch(cloud, indices[i], radius, k_indices[i], k_sqr_distances[i], max_nn);
    }
  }

 protected:
  // search.h:503-521
  void sortResults(std::vector<int> &indices, std::vector<float> &distances) const {
    std::vector<int> order(indices.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return distances[a] < distances[b]; });
    std::vector<int> si(indices.size());
    std::vector<float> sd(distances.size());
    for (size_t i = 0; i < order.size(); ++i) {
      si[i] = indices[order[i]];
      sd[i] = distances[order[i]];
    }
    indices.swap(si);
    distances.swap(sd);
  }

  PointCloudConstPtr input_;
  IndicesConstPtr indices_;
  bool sorted_results_;
  std::string name_;
};

}  // namespace search
}  // namespace pcl
#endif
