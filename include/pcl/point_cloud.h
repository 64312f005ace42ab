// pcl/point_cloud.h - pcl::PointCloud<PointT> as used on the kNN / ICP path
// (common/include/pcl/point_cloud.h).  Boost is not a dependency of this facade:
// Ptr / ConstPtr are std::shared_ptr where the reference uses boost::shared_ptr.
#ifndef PCLMI_FACADE_POINT_CLOUD_H_
#define PCLMI_FACADE_POINT_CLOUD_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pcl {

struct PCLHeader {
  std::uint32_t seq = 0;
  std::uint64_t stamp = 0;
  std::string frame_id;
};

template <typename PointT>
class PointCloud {
 public:
  typedef PointT PointType;
  typedef std::vector<PointT> VectorType;
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  typedef std::shared_ptr<const PointCloud<PointT> > ConstPtr;
  typedef typename VectorType::iterator iterator;
  typedef typename VectorType::const_iterator const_iterator;

  PointCloud() {}
  PointCloud(std::uint32_t width_, std::uint32_t height_, const PointT &value = PointT())
      : points(static_cast<size_t>(width_) * height_, value), width(width_), height(height_) {}
  // subset copy (point_cloud.h: PointCloud (const PointCloud&, const std::vector<int>&))
  PointCloud(const PointCloud<PointT> &pc, const std::vector<int> &indices)
      : header(pc.header), points(indices.size()), width(static_cast<std::uint32_t>(indices.size())), height(1),
        is_dense(pc.is_dense) {
    for (size_t i = 0; i < indices.size(); ++i) points[i] = pc.points[indices[i]];
  }

  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(size_t n) {
    points.resize(n);
    if (width * height != n) {
      width = static_cast<std::uint32_t>(n);
      height = 1;
    }
  }
  void clear() {
    points.clear();
    width = height = 0;
  }
  void push_back(const PointT &p) {
    points.push_back(p);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
  }
  bool isOrganized() const { return height > 1; }
  const PointT &operator[](size_t i) const { return points[i]; }
  PointT &operator[](size_t i) { return points[i]; }
  const PointT &at(size_t i) const { return points.at(i); }
  PointT &at(size_t i) { return points.at(i); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }

  PCLHeader header;
  VectorType points;
  std::uint32_t width = 0;
  std::uint32_t height = 0;
  bool is_dense = true;
};

typedef std::shared_ptr<std::vector<int> > IndicesPtr;
typedef std::shared_ptr<const std::vector<int> > IndicesConstPtr;

}  // namespace pcl
#endif
