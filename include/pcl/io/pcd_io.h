// pcl/io/pcd_io.h - PCD file reader / writer for the clouds either side of the kNN / ICP path
// (io/include/pcl/io/pcd_io.h; header layout io/include/pcl/io/impl/pcd_io.hpp:72-121, the
// fixtures' own headers test/bun0.pcd:1-10).  Host code only.
//
// Supported: DATA ascii and DATA binary, header versions .5 / .6 / 0.7 (COLUMNS or FIELDS, optional
// SIZE / TYPE / COUNT / VIEWPOINT), any field list - the fields named x, y, z (F 4) are read into the
// point, every other field is skipped.  DATA binary_compressed is reported as unsupported.
// Writers emit the reference's v0.7 header with FIELDS x y z, 12 bytes per point in binary files,
// exactly what PCDWriter produces for pcl::PointXYZ (impl/pcd_io.hpp:124-240,430-610).
#ifndef PCLMI_FACADE_IO_PCD_IO_H_
#define PCLMI_FACADE_IO_PCD_IO_H_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <locale>
#include <sstream>
#include <string>
#include <vector>

#include <pcl/console/print.h>
#include <pcl/point_cloud.h>

namespace pcl {
namespace io {

namespace detail {
struct PCDField {
  std::string name;
  int size = 4;
  char type = 'F';
  int count = 1;
  size_t offset = 0;  // byte offset inside one binary record
};

struct PCDHeader {
  std::vector<PCDField> fields;
  std::uint32_t width = 0, height = 1;
  size_t points = 0;
  std::string data;      // ascii / binary / binary_compressed
  size_t record = 0;     // bytes per point in a binary file
  size_t data_offset = 0;
};

// Rejects, like the reference reader (io/src/pcd_io.cpp readHeader), any header whose SIZE / TYPE / COUNT / WIDTH /
// HEIGHT / POINTS entries are missing where required, unparsable, negative or inconsistent: the binary path
// computes record offsets from them.
inline int readHeader(std::istream &f, PCDHeader &h) {
  std::string line;
  bool have_points = false, bad = false;
  // every entry of a per-field line must parse and no extra token may follow
  auto per_field = [&](std::istringstream &ss, auto setter) {
    for (auto &fd : h.fields) {
      if (!setter(ss, fd)) {
        bad = true;
        return;
      }
    }
    std::string extra;
    if (ss >> extra) bad = true;
  };
  auto read_count = [&](std::istringstream &ss, unsigned long long &v) {
    long long t = -1;
    if (!(ss >> t) || t < 0) {
      bad = true;
      return;
    }
    v = static_cast<unsigned long long>(t);
  };
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    ss.imbue(std::locale::classic());
    std::string key;
    ss >> key;
    if (key == "VERSION" || key == "VIEWPOINT") continue;
    if (key == "FIELDS" || key == "COLUMNS") {
      std::string name;
      while (ss >> name) {
        PCDField fd;
        fd.name = name;
        h.fields.push_back(fd);
      }
    } else if (key == "SIZE") {
      per_field(ss, [](std::istringstream &s, PCDField &fd) {
        return static_cast<bool>(s >> fd.size) && (fd.size == 1 || fd.size == 2 || fd.size == 4 || fd.size == 8);
      });
    } else if (key == "TYPE") {
      per_field(ss, [](std::istringstream &s, PCDField &fd) {
        return static_cast<bool>(s >> fd.type) && (fd.type == 'F' || fd.type == 'I' || fd.type == 'U');
      });
    } else if (key == "COUNT") {
      per_field(ss, [](std::istringstream &s, PCDField &fd) {
        return static_cast<bool>(s >> fd.count) && fd.count >= 1 && fd.count <= (1 << 20);
      });
    } else if (key == "WIDTH") {
      unsigned long long v = 0;
      read_count(ss, v);
      if (v > 0xffffffffull) bad = true;
      h.width = static_cast<std::uint32_t>(v);
    } else if (key == "HEIGHT") {
      unsigned long long v = 0;
      read_count(ss, v);
      if (v > 0xffffffffull) bad = true;
      h.height = static_cast<std::uint32_t>(v);
    } else if (key == "POINTS") {
      unsigned long long v = 0;
      read_count(ss, v);
      h.points = static_cast<size_t>(v);
      have_points = true;
    } else if (key == "DATA") {
      ss >> h.data;
      h.data_offset = static_cast<size_t>(f.tellg());
      break;
    }
  }
  if (bad || h.fields.empty() || h.data.empty()) return -1;
  if (!have_points) h.points = static_cast<size_t>(h.width) * h.height;
  if (h.width == 0) {  // old files without WIDTH / HEIGHT
    h.width = static_cast<std::uint32_t>(h.points);
    h.height = 1;
  }
  size_t off = 0;
  for (auto &fd : h.fields) {
    fd.offset = off;
    off += static_cast<size_t>(fd.size) * static_cast<size_t>(fd.count);  // <= 8 * 2^20 per field: no overflow
  }
  h.record = off;
  if (h.record == 0) return -1;
  return 0;
}
}  // namespace detail

/** \brief Load a PCD file into a cloud (io/include/pcl/io/pcd_io.h loadPCDFile).  Returns 0, or -1 on failure. */
template <typename PointT>
int loadPCDFile(const std::string &file_name, pcl::PointCloud<PointT> &cloud) {
  std::ifstream f(file_name.c_str(), std::ios::binary);
  if (!f) {
    PCL_ERROR("[pcl::PCDReader::read] Could not open file %s.\n", file_name.c_str());
    return -1;
  }
  detail::PCDHeader h;
  if (detail::readHeader(f, h) != 0) {
    PCL_ERROR("[pcl::PCDReader::read] %s: no valid PCD header.\n", file_name.c_str());
    return -1;
  }
  int fx = -1, fy = -1, fz = -1;
  for (size_t i = 0; i < h.fields.size(); ++i) {
    const detail::PCDField &fd = h.fields[i];
    const bool f32 = fd.type == 'F' && fd.size == 4 && fd.count == 1;
    if (fd.name == "x" && f32) fx = static_cast<int>(i);
    if (fd.name == "y" && f32) fy = static_cast<int>(i);
    if (fd.name == "z" && f32) fz = static_cast<int>(i);
  }
  if (fx < 0 || fy < 0 || fz < 0) {
    PCL_ERROR("[pcl::PCDReader::read] %s: fields x y z (F 4) not found.\n", file_name.c_str());
    return -1;
  }
  // the payload bounds the point count: a binary record is h.record bytes, an ASCII row at least two bytes per
  // field ("0 "), so a header announcing more points than the file can hold is refused before anything is allocated
  {
    f.clear();
    f.seekg(0, std::ios::end);
    const std::streamoff end = f.tellg();
    const size_t remaining = end > static_cast<std::streamoff>(h.data_offset)
                                 ? static_cast<size_t>(end) - h.data_offset : 0;
    const size_t per_point = h.data == "binary" ? h.record : 2 * h.fields.size();
    if (per_point == 0 || h.points > remaining / per_point + (h.data == "binary" ? 0 : 1)) {
      PCL_ERROR("[pcl::PCDReader::read] %s: header announces %zu points, the file holds fewer.\n", file_name.c_str(),
                h.points);
      return -1;
    }
  }
  cloud.points.assign(h.points, PointT());
  cloud.width = h.width;
  cloud.height = h.height;
  cloud.is_dense = true;
  if (static_cast<size_t>(cloud.width) * cloud.height != h.points) {
    cloud.width = static_cast<std::uint32_t>(h.points);
    cloud.height = 1;
  }
  if (h.data == "ascii") {
    f.clear();
    f.seekg(static_cast<std::streamoff>(h.data_offset));
    std::string line, tok;
    size_t n = 0;
    while (n < h.points && std::getline(f, line)) {
      if (line.empty() || line == "\r") continue;
      std::istringstream ss(line);
      ss.imbue(std::locale::classic());
      PointT &p = cloud.points[n];
      bool ok = true;
      for (size_t i = 0; i < h.fields.size() && ok; ++i)
        for (int c = 0; c < h.fields[i].count && ok; ++c) {
          if (!(ss >> tok)) {
            ok = false;
            break;
          }
          if (c != 0) continue;
          float *dst = static_cast<int>(i) == fx ? &p.x : static_cast<int>(i) == fy ? &p.y
                       : static_cast<int>(i) == fz ? &p.z : nullptr;
          if (!dst) continue;
          if (tok == "nan" || tok == "NaN" || tok == "-nan") {
            *dst = std::numeric_limits<float>::quiet_NaN();
          } else {
            std::istringstream vs(tok);
            vs.imbue(std::locale::classic());
            float v = 0.f;
            if (!(vs >> v)) v = std::numeric_limits<float>::quiet_NaN();
            *dst = v;
          }
        }
      if (!ok) {
        PCL_ERROR("[pcl::PCDReader::read] %s: short row at point %zu.\n", file_name.c_str(), n);
        return -1;
      }
      ++n;
    }
    if (n != h.points) {
      PCL_ERROR("[pcl::PCDReader::read] %s: %zu points announced, %zu read.\n", file_name.c_str(), h.points, n);
      return -1;
    }
  } else if (h.data == "binary") {
    f.clear();
    f.seekg(static_cast<std::streamoff>(h.data_offset));
    std::vector<char> raw(h.points * h.record);
    f.read(raw.data(), static_cast<std::streamsize>(raw.size()));
    if (static_cast<size_t>(f.gcount()) != raw.size()) {
      PCL_ERROR("[pcl::PCDReader::read] %s: binary payload is short.\n", file_name.c_str());
      return -1;
    }
    for (size_t i = 0; i < h.points; ++i) {
      const char *rec = raw.data() + i * h.record;
      PointT &p = cloud.points[i];
      std::memcpy(&p.x, rec + h.fields[fx].offset, 4);
      std::memcpy(&p.y, rec + h.fields[fy].offset, 4);
      std::memcpy(&p.z, rec + h.fields[fz].offset, 4);
    }
  } else {
    PCL_ERROR("[pcl::PCDReader::read] %s: DATA %s is not supported (ascii and binary are).\n", file_name.c_str(),
              h.data.c_str());
    return -1;
  }
  for (size_t i = 0; i < cloud.points.size(); ++i) {
    const PointT &p = cloud.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) {
      cloud.is_dense = false;
      break;
    }
  }
  return 0;
}

namespace detail {
template <typename PointT>
std::string generateHeader(const pcl::PointCloud<PointT> &cloud) {
  // impl/pcd_io.hpp:72-121 for a cloud with the fields x y z
  std::ostringstream oss;
  oss.imbue(std::locale::classic());
  oss << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1";
  std::uint32_t w = cloud.width, hgt = cloud.height;
  if (static_cast<size_t>(w) * hgt != cloud.points.size()) {
    w = static_cast<std::uint32_t>(cloud.points.size());
    hgt = 1;
  }
  oss << "\nWIDTH " << w << "\nHEIGHT " << hgt << "\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.points.size() << "\n";
  return oss.str();
}
}  // namespace detail

template <typename PointT>
int savePCDFileASCII(const std::string &file_name, const pcl::PointCloud<PointT> &cloud, int precision = 8) {
  if (cloud.empty()) {
    PCL_ERROR("[pcl::PCDWriter::writeASCII] Input point cloud has no data!\n");
    return -1;
  }
  std::ofstream f(file_name.c_str());
  if (!f) {
    PCL_ERROR("[pcl::PCDWriter::writeASCII] Could not open file for writing.\n");
    return -1;
  }
  f.imbue(std::locale::classic());
  f.precision(precision);
  f << detail::generateHeader(cloud) << "DATA ascii\n";
  for (size_t i = 0; i < cloud.points.size(); ++i) {
    const float v[3] = {cloud.points[i].x, cloud.points[i].y, cloud.points[i].z};
    for (int d = 0; d < 3; ++d) {
      if (std::isnan(v[d]))
        f << "nan";
      else
        f << v[d];
      f << (d < 2 ? " " : "\n");
    }
  }
  return f.good() ? 0 : -1;
}

template <typename PointT>
int savePCDFileBinary(const std::string &file_name, const pcl::PointCloud<PointT> &cloud) {
  if (cloud.empty()) {
    PCL_ERROR("[pcl::PCDWriter::writeBinary] Input point cloud has no data!\n");
    return -1;
  }
  std::ofstream f(file_name.c_str(), std::ios::binary);
  if (!f) {
    PCL_ERROR("[pcl::PCDWriter::writeBinary] Could not open file for writing.\n");
    return -1;
  }
  const std::string hdr = detail::generateHeader(cloud) + "DATA binary\n";
  f.write(hdr.data(), static_cast<std::streamsize>(hdr.size()));
  std::vector<float> raw(cloud.points.size() * 3);
  for (size_t i = 0; i < cloud.points.size(); ++i) {
    raw[3 * i + 0] = cloud.points[i].x;
    raw[3 * i + 1] = cloud.points[i].y;
    raw[3 * i + 2] = cloud.points[i].z;
  }
  f.write(reinterpret_cast<const char *>(raw.data()), static_cast<std::streamsize>(raw.size() * sizeof(float)));
  return f.good() ? 0 : -1;
}

/** \brief pcl::io::savePCDFile (cloud, binary_mode = false) */
template <typename PointT>
int savePCDFile(const std::string &file_name, const pcl::PointCloud<PointT> &cloud, bool binary_mode = false) {
  return binary_mode ? savePCDFileBinary(file_name, cloud) : savePCDFileASCII(file_name, cloud);
}

}  // namespace io
}  // namespace pcl
#endif
