// pcl_icp - chain registration of PCD files on the MI355X facade (BASELINE configs[0]'s driver).
//
//   pcl_icp [-d max_correspondence_distance] [-r ransac_threshold] [-i iterations] [--out-dir DIR] first.pcd next.pcd ...
//
// Every file after the first is aligned to the file before it with pcl::IterativeClosestPoint (RANSAC rejection on,
// the reference's defaults), the accumulated transform is applied and the moved cloud written as a binary PCD; the
// first file is written through unchanged.  The command-line CONTRACT - option letters, their defaults (0.05, 0.05,
// 50) and the lines printed per file - is that of the reference's tools/icp.cpp, so that scripts around that tool
// (and tests/test_facade_cpp.py) keep working; the program itself is this repository's own.  -n 1 selects the
// reference's Levenberg-Marquardt variant, a different algorithm outside this library's path: refused.
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include <pcl/common/transforms.h>
#include <pcl/console/parse.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_types.h>
#include <pcl/registration/icp.h>

namespace {

typedef pcl::PointCloud<pcl::PointXYZ> Cloud;

struct Options {
  double max_correspondence_distance = 0.05;  // -d
  double ransac_threshold = 0.05;             // -r
  int iterations = 50;                        // -i
  bool non_linear = false;                    // -n
  std::string out_dir;                        // --out-dir (an addition: the reference writes to the working directory)
  std::vector<std::string> files;

  static Options parse(int argc, char **argv) {
    Options o;
    pcl::console::parse_argument(argc, argv, "-d", o.max_correspondence_distance);
    pcl::console::parse_argument(argc, argv, "-r", o.ransac_threshold);
    pcl::console::parse_argument(argc, argv, "-i", o.iterations);
    pcl::console::parse_argument(argc, argv, "-n", o.non_linear);
    pcl::console::parse_argument(argc, argv, "--out-dir", o.out_dir);
    for (int at : pcl::console::parse_file_extension_argument(argc, argv, ".pcd")) o.files.push_back(argv[at]);
    return o;
  }

  std::string output_path(const std::string &input) const {
    const size_t slash = input.rfind('/');
    const std::string name = slash == std::string::npos ? input : input.substr(slash + 1);
    return out_dir.empty() ? name : out_dir + "/" + name;
  }
};

// one PCD file in memory, announced the way the reference tool announces it
bool load(const std::string &path, Cloud &cloud) {
  if (pcl::io::loadPCDFile(path, cloud) == -1) {
    std::cout << "Could not read file" << std::endl;
    return false;
  }
  std::cout << path << " width: " << cloud.width << " height: " << cloud.height << std::endl;
  return true;
}

// aligns each cloud to its predecessor and carries the product of the transforms along the chain
class Chain {
 public:
  explicit Chain(const Options &o) : opt_(o), total_(pcl::Matrix4f::Identity()) {}

  bool start(const std::string &path) {
    previous_.reset(new Cloud);
    if (!load(path, *previous_)) return false;
    const std::string out = opt_.output_path(path);
    pcl::io::savePCDFile(out, *previous_);
    std::cout << "saving first model to " << out << std::endl;
    return true;
  }

  bool append(const std::string &path) {
    Cloud::Ptr current(new Cloud);
    if (!load(path, *current)) return false;
    std::cout << "Using IterativeClosestPoint" << std::endl;
    pcl::IterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> icp;
    icp.setMaxCorrespondenceDistance(opt_.max_correspondence_distance);
    icp.setRANSACOutlierRejectionThreshold(opt_.ransac_threshold);
    icp.setMaximumIterations(opt_.iterations);
    icp.setInputCloud(current);
    icp.setInputTarget(previous_);
    Cloud aligned;
    icp.align(aligned);
    const pcl::Matrix4f step = icp.getFinalTransformation();
    std::cout << step << std::endl;
    total_ = step * total_;
    Cloud moved;
    pcl::transformPointCloud(*current, moved, total_);
    const std::string out = opt_.output_path(path);
    pcl::io::savePCDFileBinary(out, moved);
    std::cout << "saving result to " << out << std::endl;
    previous_ = current;  // the next file is registered to this one as it was read
    return true;
  }

 private:
  const Options &opt_;
  Cloud::Ptr previous_;
  pcl::Matrix4f total_;
};

}  // namespace

int main(int argc, char **argv) {
  const Options opt = Options::parse(argc, argv);
  if (opt.non_linear) {
    std::cout << "IterativeClosestPointNonLinear is not part of this library (point-to-point ICP only)" << std::endl;
    return -1;
  }
  if (opt.files.empty()) {
    std::cout << "usage: pcl_icp [-d dist] [-r ransac_thr] [-i iterations] [--out-dir dir] first.pcd next.pcd ..."
              << std::endl;
    return -1;
  }
  Chain chain(opt);
  if (!chain.start(opt.files.front())) return -1;
  for (size_t i = 1; i < opt.files.size(); ++i)
    if (!chain.append(opt.files[i])) return -1;
  return 0;
}
