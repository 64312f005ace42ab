// pcl/search/kdtree.h - pcl::search::KdTree<PointT> (search/include/pcl/search/kdtree.h:58-174):
// the search facade over KdTreeFLANN.  Unlike the reference, whose batched overloads are serial loops
// over the single-query call (search.h:244-262), the batched overloads here hand the whole batch to
// the device in one call.
#ifndef PCLMI_FACADE_SEARCH_KDTREE_H_
#define PCLMI_FACADE_SEARCH_KDTREE_H_

#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/search/search.h>

namespace pcl {
namespace search {

template <typename PointT>
class KdTree : public Search<PointT> {
 public:
  typedef typename Search<PointT>::PointCloud PointCloud;
  typedef typename Search<PointT>::PointCloudConstPtr PointCloudConstPtr;
  typedef pcl::IndicesConstPtr IndicesConstPtr;
  using Search<PointT>::input_;
  using Search<PointT>::indices_;
  using Search<PointT>::sorted_results_;
  using Search<PointT>::nearestKSearch;
  using Search<PointT>::radiusSearch;

  typedef std::shared_ptr<KdTree<PointT> > Ptr;
  typedef std::shared_ptr<const KdTree<PointT> > ConstPtr;
  typedef std::shared_ptr<pcl::KdTreeFLANN<PointT> > KdTreeFLANNPtr;
  typedef std::shared_ptr<const pcl::KdTreeFLANN<PointT> > KdTreeFLANNConstPtr;
  typedef std::shared_ptr<const pcl::PointRepresentation<PointT> > PointRepresentationConstPtr;

  explicit KdTree(bool sorted = true) : Search<PointT>("KdTree", sorted), tree_(new pcl::KdTreeFLANN<PointT>(sorted)) {}
  virtual ~KdTree() {}

  void setPointRepresentation(const PointRepresentationConstPtr &point_representation) {
    tree_->setPointRepresentation(point_representation);
  }
  PointRepresentationConstPtr getPointRepresentation() const { return tree_->getPointRepresentation(); }

  void setSortedResults(bool sorted_results) override {  // kdtree.h:104-109
    sorted_results_ = sorted_results;
    tree_->setSortedResults(sorted_results);
  }
  void setEpsilon(float eps) { tree_->setEpsilon(eps); }
  float getEpsilon() const { return tree_->getEpsilon(); }

  void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr &indices = IndicesConstPtr()) override {
    tree_->setInputCloud(cloud, indices);
    input_ = cloud;
    indices_ = indices;
  }

  int nearestKSearch(const PointT &point, int k, std::vector<int> &k_indices,
                     std::vector<float> &k_sqr_distances) const override {
    return tree_->nearestKSearch(point, k, k_indices, k_sqr_distances);
  }

  int radiusSearch(const PointT &point, double radius, std::vector<int> &k_indices,
                   std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const override {
    return tree_->radiusSearch(point, radius, k_indices, k_sqr_distances, max_nn);
  }

  // batched kNN: one device call
  void nearestKSearch(const PointCloud &cloud, const std::vector<int> &indices, int k,
                      std::vector<std::vector<int> > &k_indices,
                      std::vector<std::vector<float> > &k_sqr_distances) const override {
    std::vector<float> q;
    const size_t nq = gatherQueries(cloud, indices, q);
    std::vector<int> flat_i;
    std::vector<float> flat_d;
    const int ke = tree_->nearestKSearchBatch(q.data(), nq, 12, k, flat_i, flat_d);
    k_indices.assign(nq, std::vector<int>());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (ke <= 0) return;
    for (size_t i = 0; i < nq; ++i) {
      k_indices[i].assign(flat_i.begin() + i * ke, flat_i.begin() + (i + 1) * ke);
      k_sqr_distances[i].assign(flat_d.begin() + i * ke, flat_d.begin() + (i + 1) * ke);
    }
  }

  // batched radius search: one device call
  void radiusSearch(const PointCloud &cloud, const std::vector<int> &indices, double radius,
                    std::vector<std::vector<int> > &k_indices, std::vector<std::vector<float> > &k_sqr_distances,
                    unsigned int max_nn = 0) const override {
    std::vector<float> q;
    const size_t nq = gatherQueries(cloud, indices, q);
    std::vector<uint64_t> off;
    std::vector<int> flat_i;
    std::vector<float> flat_d;
    tree_->radiusSearchBatch(q.data(), nq, 12, radius, off, flat_i, flat_d, max_nn);
    k_indices.assign(nq, std::vector<int>());
    k_sqr_distances.assign(nq, std::vector<float>());
    for (size_t i = 0; i < nq; ++i) {
      k_indices[i].assign(flat_i.begin() + off[i], flat_i.begin() + off[i + 1]);
      k_sqr_distances[i].assign(flat_d.begin() + off[i], flat_d.begin() + off[i + 1]);
    }
  }

  KdTreeFLANNPtr getKdTree() const { return tree_; }

 protected:
  size_t gatherQueries(const PointCloud &cloud, const std::vector<int> &indices, std::vector<float> &q) const {
    const size_t nq = indices.empty() ? cloud.size() : indices.size();
    q.resize(nq * 3);
    for (size_t i = 0; i < nq; ++i) tree_->toXYZ(cloud.points[indices.empty() ? i : indices[i]], &q[i * 3]);
    return nq;
  }

  KdTreeFLANNPtr tree_;
};

}  // namespace search
}  // namespace pcl
#endif
