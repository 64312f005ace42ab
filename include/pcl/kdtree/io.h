// pcl/kdtree/io.h - pcl::getApproximateIndices (kdtree/include/pcl/kdtree/io.h:47-68, impl/io.hpp:47-84): for
// every point of cloud_in the index of its nearest point in cloud_ref.  The reference builds a KdTreeFLANN on
// cloud_ref and queries it point by point; here the whole cloud goes through ONE batched 1-NN call.
#ifndef PCLMI_FACADE_KDTREE_IO_H_
#define PCLMI_FACADE_KDTREE_IO_H_

#include <pcl/kdtree/kdtree_flann.h>

#include <vector>

namespace pcl {

template <typename Point1T, typename Point2T>
void getApproximateIndices(const typename pcl::PointCloud<Point1T>::Ptr &cloud_in,
                           const typename pcl::PointCloud<Point2T>::Ptr &cloud_ref, std::vector<int> &indices) {
  indices.assign(cloud_in->points.size(), 0);
  if (cloud_in->points.empty()) return;
  pcl::KdTreeFLANN<Point2T> tree;
  tree.setInputCloud(cloud_ref);
  const pclmi_tree *t = tree.getDeviceTree();
  if (!t || tree.getNumberOfIndexedPoints() == 0) return;  // empty reference cloud (the reference reads nn_idx[0] of a failed search)
  std::vector<float> dists(cloud_in->points.size());
  int k_eff = 0;
  const int rc = pclmi_knn(t, &cloud_in->points[0].x, cloud_in->points.size(), sizeof(Point1T), 1, indices.data(),
                           dists.data(), &k_eff);
  if (rc != PCLMI_OK) PCL_ERROR("[pcl::getApproximateIndices] %s\n", pclmi_last_error());
}

template <typename PointT>
void getApproximateIndices(const typename pcl::PointCloud<PointT>::Ptr &cloud_in,
                           const typename pcl::PointCloud<PointT>::Ptr &cloud_ref, std::vector<int> &indices) {
  getApproximateIndices<PointT, PointT>(cloud_in, cloud_ref, indices);
}

}  // namespace pcl
#endif
