// pcl_icp - the reference's tools/icp.cpp (tools/icp.cpp:56-138) against the MI355X facade: registers
// every further .pcd file of the command line to the one before it with pcl::IterativeClosestPoint,
// prints the transforms and writes the moved clouds next to the working directory.
//   pcl_icp [-d max_correspondence_distance=0.05] [-r ransac_threshold=0.05] [-i iterations=50] a.pcd b.pcd ...
// Options follow the reference; -n 1 (IterativeClosestPointNonLinear, a Levenberg-Marquardt solver) is a
// different algorithm outside this library's path and is refused.  --out-dir DIR (an addition) chooses
// where results are written instead of the working directory.
// Attribution: the command line, the option letters and the per-file loop of this tool follow the reference's
// tools/icp.cpp so that configs[0]'s driver can be run unchanged against the facade.  That file carries the notice
//
//   Software License Agreement (BSD License)
//   Point Cloud Library (PCL) - www.pointclouds.org
//   Copyright (c) 2011, Willow Garage, Inc.  All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that
//   the following conditions are met:
//    * Redistributions of source code must retain the above copyright notice, this list of conditions and the
//      following disclaimer.
//    * Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the
//      following disclaimer in the documentation and/or other materials provided with the distribution.
//    * Neither the name of the copyright holder(s) nor the names of its contributors may be used to endorse or
//      promote products derived from this software without specific prior written permission.
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED
//   WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A
//   PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY
//   DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT LIMITED TO,
//   PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER
//   CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR
//   OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
//
// Everything the tool calls (pcl::IterativeClosestPoint, the PCD reader / writer, the option parser) is this
// repository's own code behind the reference's names.
#include <iostream>
#include <string>
#include <vector>

#include <pcl/common/transforms.h>
#include <pcl/console/parse.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_types.h>
#include <pcl/registration/icp.h>

typedef pcl::PointXYZ PointType;
typedef pcl::PointCloud<PointType> Cloud;
typedef Cloud::Ptr CloudPtr;

static std::string result_name(const std::string &dir, const std::string &path) {
  const std::string base = path.substr(path.rfind("/") + 1);
  return dir.empty() ? base : dir + "/" + base;
}

int main(int argc, char **argv) {
  double dist = 0.05;
  pcl::console::parse_argument(argc, argv, "-d", dist);
  double rans = 0.05;
  pcl::console::parse_argument(argc, argv, "-r", rans);
  int iter = 50;
  pcl::console::parse_argument(argc, argv, "-i", iter);
  bool nonLinear = false;
  pcl::console::parse_argument(argc, argv, "-n", nonLinear);
  std::string out_dir;
  pcl::console::parse_argument(argc, argv, "--out-dir", out_dir);
  if (nonLinear) {
    std::cout << "IterativeClosestPointNonLinear is not part of this library (point-to-point ICP only)" << std::endl;
    return -1;
  }
  std::vector<int> pcd_indices = pcl::console::parse_file_extension_argument(argc, argv, ".pcd");
  if (pcd_indices.empty()) {
    std::cout << "usage: pcl_icp [-d dist] [-r ransac_thr] [-i iterations] model.pcd data.pcd ..." << std::endl;
    return -1;
  }

  CloudPtr model(new Cloud);
  if (pcl::io::loadPCDFile(argv[pcd_indices[0]], *model) == -1) {
    std::cout << "Could not read file" << std::endl;
    return -1;
  }
  std::cout << argv[pcd_indices[0]] << " width: " << model->width << " height: " << model->height << std::endl;
  std::string result_filename = result_name(out_dir, argv[pcd_indices[0]]);
  pcl::io::savePCDFile(result_filename, *model);
  std::cout << "saving first model to " << result_filename << std::endl;

  pcl::Matrix4f t(pcl::Matrix4f::Identity());
  for (size_t i = 1; i < pcd_indices.size(); i++) {
    CloudPtr data(new Cloud);
    if (pcl::io::loadPCDFile(argv[pcd_indices[i]], *data) == -1) {
      std::cout << "Could not read file" << std::endl;
      return -1;
    }
    std::cout << argv[pcd_indices[i]] << " width: " << data->width << " height: " << data->height << std::endl;
    std::cout << "Using IterativeClosestPoint" << std::endl;
    pcl::IterativeClosestPoint<PointType, PointType> icp;
    icp.setMaximumIterations(iter);
    icp.setMaxCorrespondenceDistance(dist);
    icp.setRANSACOutlierRejectionThreshold(rans);
    icp.setInputTarget(model);
    icp.setInputCloud(data);
    CloudPtr tmp(new Cloud);
    icp.align(*tmp);
    t = icp.getFinalTransformation() * t;
    pcl::transformPointCloud(*data, *tmp, t);
    std::cout << icp.getFinalTransformation() << std::endl;
    *model = *data;
    result_filename = result_name(out_dir, argv[pcd_indices[i]]);
    pcl::io::savePCDFileBinary(result_filename, *tmp);
    std::cout << "saving result to " << result_filename << std::endl;
  }
  return 0;
}
