// pcl/kdtree/kdtree_flann.h - pcl::KdTreeFLANN<PointT> on the MI355X library.
//
// Same class name and methods as kdtree/include/pcl/kdtree/kdtree_flann.h:56-249; the FLANN index
// behind it is replaced by a device-resident tree (pclmi_tree, include/pclmi.h).  Exact search
// (eps = 0), results ascending by (squared distance, index).  Differences a caller can observe:
//  - copies of a KdTreeFLANN share one device tree (shared ownership) instead of the reference's
//    shallow raw-pointer copy (kdtree_flann.h:112-125);
//  - setEpsilon is accepted and ignored: the search is always exact;
//  - additional batched entry points nearestKSearchBatch / radiusSearchBatch: ONE library call for
//    many queries - the form in which the GPU is worth using.  The per-point methods work, each
//    costing one host<->device round trip.
#ifndef PCLMI_FACADE_KDTREE_FLANN_H_
#define PCLMI_FACADE_KDTREE_FLANN_H_

#include <cmath>
#include <limits>

#include <pcl/console/print.h>
#include <pcl/kdtree/kdtree.h>
#include <pclmi.h>

namespace pcl {

template <typename PointT>
class KdTreeFLANN : public pcl::KdTree<PointT> {
 public:
  using KdTree<PointT>::input_;
  using KdTree<PointT>::indices_;
  using KdTree<PointT>::epsilon_;
  using KdTree<PointT>::sorted_;
  using KdTree<PointT>::point_representation_;
  using KdTree<PointT>::nearestKSearch;
  using KdTree<PointT>::radiusSearch;

  typedef typename KdTree<PointT>::PointCloud PointCloud;
  typedef typename KdTree<PointT>::PointCloudConstPtr PointCloudConstPtr;
  typedef std::shared_ptr<KdTreeFLANN<PointT> > Ptr;
  typedef std::shared_ptr<const KdTreeFLANN<PointT> > ConstPtr;

  explicit KdTreeFLANN(bool sorted = true) : pcl::KdTree<PointT>(sorted), dim_(0), total_nr_points_(0) {}
  virtual ~KdTreeFLANN() {}

  void setEpsilon(float eps) override { epsilon_ = eps; }  // accepted; the search stays exact
  void setSortedResults(bool sorted) { sorted_ = sorted; }  // kdtree_flann.h:138-144
  Ptr makeShared() { return Ptr(new KdTreeFLANN<PointT>(*this)); }

  // impl/kdtree_flann.hpp:46-75
  void setInputCloud(const PointCloudConstPtr &cloud, const IndicesConstPtr &indices = IndicesConstPtr()) override {
    tree_.reset();
    total_nr_points_ = 0;
    epsilon_ = 0.0f;
    dim_ = point_representation_->getNumberOfDimensions();
    input_ = cloud;
    indices_ = indices;
    if (!input_) return;
    if (dim_ < 1 || dim_ > 3) {
      PCL_ERROR("[pcl::KdTreeFLANN::setInputCloud] %d-dimensional point representations are not supported (1..3)\n",
                dim_);
      return;
    }
    const size_t n = input_->points.size();
    const int32_t *idx = indices_ ? reinterpret_cast<const int32_t *>(indices_->data()) : nullptr;
    const size_t n_idx = indices_ ? indices_->size() : 0;
    pclmi_tree *t = nullptr;
    int rc;
    if (point_representation_->isTrivial()) {
      rc = pclmi_tree_create(n ? reinterpret_cast<const float *>(&input_->points[0]) : nullptr, n, sizeof(PointT), idx,
                             n_idx, 15, &t);
    } else {
      // convertCloudToArray (:180-249): vectorize through the representation; invalid rows become
      // NaN rows, which the library skips exactly like the reference's isValid test (:199-203)
      std::vector<float> packed(n * 3, 0.f);
      for (size_t i = 0; i < n; ++i) toXYZ(input_->points[i], &packed[i * 3]);
      rc = pclmi_tree_create(packed.data(), n, 12, idx, n_idx, 15, &t);
    }
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::KdTreeFLANN::setInputCloud] %s\n", pclmi_last_error());
      return;
    }
    tree_.reset(new Handle(t));
    total_nr_points_ = static_cast<int>(pclmi_tree_size(t));
    if (total_nr_points_ == 0)
      PCL_ERROR("[pcl::KdTreeFLANN::setInputCloud] Cannot create a KDTree with an empty input cloud!\n");  // :57-61
  }

  // impl/kdtree_flann.hpp:78-112
  int nearestKSearch(const PointT &point, int k, std::vector<int> &k_indices,
                     std::vector<float> &k_sqr_distances) const override {
    assert(point_representation_->isValid(point) && "Invalid (NaN, Inf) point coordinates given to nearestKSearch!");
    if (k > total_nr_points_) k = total_nr_points_;
    k_indices.resize(k > 0 ? k : 0);
    k_sqr_distances.resize(k > 0 ? k : 0);
    if (k <= 0 || !tree_) return 0;
    float q[3];
    toXYZ(point, q);
    int k_eff = 0;
    const int rc = pclmi_knn(tree_->t, q, 1, 12, k, k_indices.data(), k_sqr_distances.data(), &k_eff);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::KdTreeFLANN::nearestKSearch] %s\n", pclmi_last_error());
      k_indices.clear();
      k_sqr_distances.clear();
      return 0;
    }
    return k_eff;
  }

  // impl/kdtree_flann.hpp:115-157
  int radiusSearch(const PointT &point, double radius, std::vector<int> &k_indices,
                   std::vector<float> &k_sqr_distances, unsigned int max_nn = 0) const override {
    assert(point_representation_->isValid(point) && "Invalid (NaN, Inf) point coordinates given to radiusSearch!");
    k_indices.clear();
    k_sqr_distances.clear();
    if (!tree_) return 0;
    float q[3];
    toXYZ(point, q);
    uint64_t offsets[2] = {0, 0};
    int32_t *idx = nullptr;
    float *d2 = nullptr;
    const int rc = pclmi_radius(tree_->t, q, 1, 12, radius, max_nn, sorted_ ? 1 : 0, offsets, &idx, &d2);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::KdTreeFLANN::radiusSearch] %s\n", pclmi_last_error());
      return 0;
    }
    k_indices.assign(idx, idx + offsets[1]);
    k_sqr_distances.assign(d2, d2 + offsets[1]);
    pclmi_free(idx);
    pclmi_free(d2);
    return static_cast<int>(offsets[1]);
  }

  // ---- batched extensions (one library call for all queries) -----------------------------------
  // queries: nq rows of 3 floats at `stride_bytes`, already in the representation's space.
  // Outputs are nq x k_eff row-major; returns k_eff (k clamped to the tree size).
  int nearestKSearchBatch(const float *queries, size_t nq, size_t stride_bytes, int k, std::vector<int> &indices,
                          std::vector<float> &sqr_distances) const {
    if (k > total_nr_points_) k = total_nr_points_;
    if (k <= 0 || !tree_) {
      indices.clear();
      sqr_distances.clear();
      return 0;
    }
    indices.resize(nq * static_cast<size_t>(k));
    sqr_distances.resize(nq * static_cast<size_t>(k));
    int k_eff = 0;
    const int rc = pclmi_knn(tree_->t, queries, nq, stride_bytes, k, indices.data(), sqr_distances.data(), &k_eff);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::KdTreeFLANN::nearestKSearchBatch] %s\n", pclmi_last_error());
      return 0;
    }
    return k_eff;
  }

  // CSR output: offsets has nq + 1 entries.  Returns the total number of neighbours.
  size_t radiusSearchBatch(const float *queries, size_t nq, size_t stride_bytes, double radius,
                           std::vector<uint64_t> &offsets, std::vector<int> &indices,
                           std::vector<float> &sqr_distances, unsigned int max_nn = 0) const {
    offsets.assign(nq + 1, 0);
    indices.clear();
    sqr_distances.clear();
    if (!tree_) return 0;
    int32_t *idx = nullptr;
    float *d2 = nullptr;
    const int rc = pclmi_radius(tree_->t, queries, nq, stride_bytes, radius, max_nn, sorted_ ? 1 : 0, offsets.data(),
                                &idx, &d2);
    if (rc != PCLMI_OK) {
      PCL_ERROR("[pcl::KdTreeFLANN::radiusSearchBatch] %s\n", pclmi_last_error());
      return 0;
    }
    indices.assign(idx, idx + offsets[nq]);
    sqr_distances.assign(d2, d2 + offsets[nq]);
    pclmi_free(idx);
    pclmi_free(d2);
    return static_cast<size_t>(offsets[nq]);
  }

  // the representation's view of a point, padded to three coordinates; NaN when invalid
  void toXYZ(const PointT &p, float out[3]) const {
    out[0] = out[1] = out[2] = 0.f;
    if (!point_representation_->isValid(p)) {
      out[0] = out[1] = out[2] = std::numeric_limits<float>::quiet_NaN();
      return;
    }
    point_representation_->vectorize(p, out);
  }

  const pclmi_tree *getDeviceTree() const { return tree_ ? tree_->t : nullptr; }
  int getNumberOfIndexedPoints() const { return total_nr_points_; }

 private:
  struct Handle {
    pclmi_tree *t;
    explicit Handle(pclmi_tree *tt) : t(tt) {}
    ~Handle() { pclmi_tree_destroy(t); }
    Handle(const Handle &) = delete;
    Handle &operator=(const Handle &) = delete;
  };
  std::string getName() const override { return "KdTreeFLANN"; }

  std::shared_ptr<Handle> tree_;
  int dim_;
  int total_nr_points_;
};

}  // namespace pcl
#endif
