// pcl/features/feature.h - pcl::Feature<PointInT, PointOutT> as far as the normal-estimation path needs it
// (features/include/pcl/features/feature.h:100-250, impl/feature.hpp:91-220): input cloud + index view,
// search surface, search method, k / radius.  The reference binds the per-point nearestKSearch / radiusSearch
// of the search method into a functor and loops (impl/feature.hpp:143-159); here derived classes hand the whole
// index list to the device in one call, so the search method only has to provide the surface's device tree.
#ifndef PCLMI_FACADE_FEATURES_FEATURE_H_
#define PCLMI_FACADE_FEATURES_FEATURE_H_

#include <memory>
#include <string>

#include <pcl/console/print.h>
#include <pcl/pcl_base.h>
#include <pcl/search/kdtree.h>

namespace pcl {

template <typename PointInT, typename PointOutT>
class Feature : public PCLBase<PointInT> {
 public:
  using PCLBase<PointInT>::indices_;
  using PCLBase<PointInT>::input_;
  typedef PCLBase<PointInT> BaseClass;
  typedef std::shared_ptr<Feature<PointInT, PointOutT> > Ptr;
  typedef pcl::search::Search<PointInT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef pcl::PointCloud<PointInT> PointCloudIn;
  typedef typename PointCloudIn::ConstPtr PointCloudInConstPtr;
  typedef pcl::PointCloud<PointOutT> PointCloudOut;

  Feature() : feature_name_(), search_parameter_(0), search_radius_(0), k_(0), fake_surface_(false) {}
  virtual ~Feature() {}

  void setSearchSurface(const PointCloudInConstPtr &cloud) {
    surface_ = cloud;
    fake_surface_ = false;
  }
  PointCloudInConstPtr getSearchSurface() const { return surface_; }
  void setSearchMethod(const KdTreePtr &tree) { tree_ = tree; }
  KdTreePtr getSearchMethod() const { return tree_; }
  double getSearchParameter() const { return search_parameter_; }
  void setKSearch(int k) { k_ = k; }
  int getKSearch() const { return k_; }
  void setRadiusSearch(double radius) { search_radius_ = radius; }
  double getRadiusSearch() const { return search_radius_; }

  /** \brief impl/feature.hpp:188-220 */
  void compute(PointCloudOut &output) {
    if (!initCompute()) {
      output.width = output.height = 0;
      output.points.clear();
      return;
    }
    output.header = input_->header;
    if (output.points.size() != indices_->size()) output.points.resize(indices_->size());
    if (indices_->size() != input_->points.size()) {
      output.width = static_cast<std::uint32_t>(indices_->size());
      output.height = 1;
    } else {
      output.width = input_->width;
      output.height = input_->height;
    }
    output.is_dense = input_->is_dense;
    computeFeature(output);
    deinitCompute();
  }

 protected:
  /** \brief impl/feature.hpp:91-169 */
  virtual bool initCompute() {
    if (!PCLBase<PointInT>::initCompute()) {
      PCL_ERROR("[pcl::%s::initCompute] Init failed.\n", getClassName().c_str());
      return false;
    }
    if (input_->points.empty()) {
      PCL_ERROR("[pcl::%s::compute] input_ is empty!\n", getClassName().c_str());
      deinitCompute();
      return false;
    }
    if (!surface_) {
      fake_surface_ = true;
      surface_ = input_;
    }
    if (!tree_) tree_.reset(new pcl::search::KdTree<PointInT>(false));
    if (tree_->getInputCloud() != surface_) tree_->setInputCloud(surface_);  // the tree searches the surface
    if (search_radius_ != 0.0) {
      if (k_ != 0) {
        PCL_ERROR("[pcl::%s::compute] Both radius (%f) and K (%d) defined! Set one of them to zero first and then "
                  "re-run compute ().\n", getClassName().c_str(), search_radius_, k_);
        deinitCompute();
        return false;
      }
      search_parameter_ = search_radius_;
    } else if (k_ != 0) {
      search_parameter_ = k_;
    } else {
      PCL_ERROR("[pcl::%s::compute] Neither radius nor K defined! Set one of them to a positive number first and "
                "then re-run compute ().\n", getClassName().c_str());
      deinitCompute();
      return false;
    }
    return true;
  }
  virtual bool deinitCompute() {
    if (fake_surface_) {
      surface_.reset();
      fake_surface_ = false;
    }
    return true;
  }
  const std::string &getClassName() const { return feature_name_; }
  virtual void computeFeature(PointCloudOut &output) = 0;

  /** \brief the device tree of the search surface: the search method's own when it is a search::KdTree
   *  (any other Search implementation gets a private tree over the same surface) */
  const pclmi_tree *surfaceDeviceTree() {
    if (auto kd = std::dynamic_pointer_cast<pcl::search::KdTree<PointInT> >(tree_))
      if (kd->getKdTree() && kd->getKdTree()->getDeviceTree()) return kd->getKdTree()->getDeviceTree();
    if (!own_tree_ || own_tree_->getInputCloud() != surface_) {
      own_tree_.reset(new pcl::KdTreeFLANN<PointInT>(false));
      own_tree_->setInputCloud(surface_);
    }
    return own_tree_->getDeviceTree();
  }

  std::string feature_name_;
  KdTreePtr tree_;
  PointCloudInConstPtr surface_;
  double search_parameter_;
  double search_radius_;
  int k_;
  bool fake_surface_;
  std::shared_ptr<pcl::KdTreeFLANN<PointInT> > own_tree_;
};

}  // namespace pcl
#endif
