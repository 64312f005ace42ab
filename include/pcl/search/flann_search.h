// pcl/search/flann_search.h - pcl::search::FlannSearch<PointT> (search/include/pcl/search/flann_search.h:
// 60-200, impl/flann_search.hpp:51-338): the reference's generic FLANN wrapper whose batched overloads hand
// the whole query matrix to the index in one call (impl/flann_search.hpp:126-209).  Here the index behind it
// is the same device-resident tree as search::KdTree; batched calls are one pclmi_knn / pclmi_radius each.
//
// Differences from the reference: the FlannIndexCreator hierarchy only selects FLANN's index type in the
// reference - KdTreeIndexCreator (the default, exact single kd-tree, leaf size 15) is the one this library
// implements, so the creator objects are accepted and carry no behaviour; setEpsilon is accepted and ignored
// (the search is always exact).
#ifndef PCLMI_FACADE_SEARCH_FLANN_SEARCH_H_
#define PCLMI_FACADE_SEARCH_FLANN_SEARCH_H_

#include <memory>

#include <pcl/search/kdtree.h>

namespace pcl {
namespace search {

template <typename PointT>
class FlannSearch : public KdTree<PointT> {
 public:
  typedef std::shared_ptr<FlannSearch<PointT> > Ptr;
  typedef std::shared_ptr<const FlannSearch<PointT> > ConstPtr;

  class FlannIndexCreator {
   public:
    virtual ~FlannIndexCreator() {}
  };
  class KdTreeIndexCreator : public FlannIndexCreator {
   public:
    explicit KdTreeIndexCreator(unsigned int max_leaf_size = 15) : max_leaf_size_(max_leaf_size) {}
    unsigned int getMaxLeafSize() const { return max_leaf_size_; }

   private:
    unsigned int max_leaf_size_;
  };

  /** \brief takes ownership of `creator`, as the reference does (flann_search.h:129, impl/flann_search.hpp:64) */
  explicit FlannSearch(bool sorted = true, FlannIndexCreator *creator = new KdTreeIndexCreator())
      : KdTree<PointT>(sorted), creator_(creator), eps_(0.0) {
    this->name_ = "FlannSearch";
  }
  virtual ~FlannSearch() {}

  void setEpsilon(double eps) { eps_ = eps; }
  double getEpsilon() { return eps_; }

 private:
  std::unique_ptr<FlannIndexCreator> creator_;
  double eps_;
};

}  // namespace search
}  // namespace pcl
#endif
