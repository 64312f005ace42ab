// pcl/io/pcd_io.h - PCD file reader / writer for the clouds either side of the kNN / ICP path
// (io/include/pcl/io/pcd_io.h; header layout io/include/pcl/io/impl/pcd_io.hpp:72-121, the
// fixtures' own headers test/bun0.pcd:1-10).  Host code only.
//
// Supported: DATA ascii, DATA binary and DATA binary_compressed (io/src/pcd_io.cpp:1116-1193: two 32-bit sizes,
// then the LZF-compressed field-major image of the cloud - all x, then all y ...; padding fields named "_" are not
// stored), header versions .5 / .6 / 0.7 (COLUMNS or FIELDS, optional SIZE / TYPE / COUNT / VIEWPOINT), any field
// list - the fields named x, y, z (F 4) are read into the point, every other field is skipped.
// Writers emit the reference's v0.7 header with FIELDS x y z, 12 bytes per point in binary files,
// exactly what PCDWriter produces for pcl::PointXYZ (impl/pcd_io.hpp:124-240,430-610).
#ifndef PCLMI_FACADE_IO_PCD_IO_H_
#define PCLMI_FACADE_IO_PCD_IO_H_

#include <algorithm>
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

namespace detail {
// LZF (Marc Lehmann's liblzf byte format, what pcl::lzfDecompress / lzfCompress read and write, io/src/lzf.cpp):
// a control byte below 32 starts a literal run of ctrl + 1 bytes; otherwise it is a back reference of length
// (ctrl >> 5) + 2 (a length code of 7 is extended by the next byte) at distance ((ctrl & 31) << 8 | next byte) + 1.
// Returns the number of bytes produced, 0 on malformed input or when the output does not fit.
inline size_t lzfDecompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return 0;
      std::memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (ip >= in_len) return 0;
      if (len == 7) {
        len += in[ip++];
        if (ip >= in_len) return 0;
      }
      const size_t dist = ((static_cast<size_t>(ctrl & 31u)) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return 0;
      for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return op;
}

// Greedy LZF compressor (3-byte hash, most recent occurrence).  out must hold in_len + in_len / 16 + 64 bytes.
inline size_t lzfCompress(const unsigned char *in, size_t in_len, unsigned char *out) {
  std::vector<std::uint32_t> table(1u << 16, 0xffffffffu);
  size_t ip = 0, op = 0, lit = 0;  // lit: start of the pending literal run
  auto flush_literals = [&](size_t upto) {
    while (lit < upto) {
      const size_t run = std::min<size_t>(32, upto - lit);
      out[op++] = static_cast<unsigned char>(run - 1);
      std::memcpy(out + op, in + lit, run);
      op += run;
      lit += run;
    }
  };
  while (ip + 2 < in_len) {
    const std::uint32_t v = static_cast<std::uint32_t>(in[ip]) << 16 | static_cast<std::uint32_t>(in[ip + 1]) << 8 | in[ip + 2];
    const std::uint32_t hsh = ((v * 2654435761u) >> 16) & 0xffffu;
    const std::uint32_t ref = table[hsh];
    table[hsh] = static_cast<std::uint32_t>(ip);
    if (ref != 0xffffffffu && ip - ref <= 8192 && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] &&
        in[ref + 2] == in[ip + 2]) {
      size_t len = 3;
      while (ip + len < in_len && len < 264 && in[ref + len] == in[ip + len]) ++len;
      flush_literals(ip);
      const size_t dist = ip - ref - 1, l = len - 2;
      if (l < 7) {
        out[op++] = static_cast<unsigned char>((l << 5) | (dist >> 8));
      } else {
        out[op++] = static_cast<unsigned char>((7u << 5) | (dist >> 8));
        out[op++] = static_cast<unsigned char>(l - 7);
      }
      out[op++] = static_cast<unsigned char>(dist & 0xff);
      ip += len;
      lit = ip;
    } else {
      ++ip;
    }
  }
  flush_literals(in_len);
  return op;
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
    if (h.data == "binary_compressed") {
      // an LZF stream expands at most 264 / 3 = 88-fold (the longest back reference takes three bytes)
      if (remaining < 8 || h.points > (remaining - 8) * 88 + 1) {
        PCL_ERROR("[pcl::PCDReader::read] %s: header announces %zu points, the compressed payload cannot hold them.\n",
                  file_name.c_str(), h.points);
        return -1;
      }
    } else if (per_point == 0 || h.points > remaining / per_point + (h.data == "binary" ? 0 : 1)) {
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
  } else if (h.data == "binary_compressed") {
    // io/src/pcd_io.cpp:1116-1193
    f.clear();
    f.seekg(0, std::ios::end);
    const size_t file_end = static_cast<size_t>(f.tellg());
    f.seekg(static_cast<std::streamoff>(h.data_offset));
    std::uint32_t sizes[2] = {0, 0};  // compressed, uncompressed
    f.read(reinterpret_cast<char *>(sizes), 8);
    if (f.gcount() != 8 || sizes[0] > file_end - h.data_offset - 8) {
      PCL_ERROR("[pcl::PCDReader::read] %s: compressed payload is short.\n", file_name.c_str());
      return -1;
    }
    // field-major layout: one block per field that is not padding
    size_t stored = 0, off_x = 0, off_y = 0, off_z = 0;
    for (size_t i = 0; i < h.fields.size(); ++i) {
      const detail::PCDField &fd = h.fields[i];
      if (fd.name == "_") continue;
      if (static_cast<int>(i) == fx) off_x = stored;
      if (static_cast<int>(i) == fy) off_y = stored;
      if (static_cast<int>(i) == fz) off_z = stored;
      stored += static_cast<size_t>(fd.size) * static_cast<size_t>(fd.count) * h.points;
    }
    if (stored != sizes[1]) {
      PCL_ERROR("[pcl::PCDReader::read] %s: the header describes %zu bytes, the compressed block says %u.\n",
                file_name.c_str(), stored, sizes[1]);
      return -1;
    }
    std::vector<unsigned char> packed(sizes[0]), raw(stored ? stored : 1);
    f.read(reinterpret_cast<char *>(packed.data()), static_cast<std::streamsize>(packed.size()));
    if (static_cast<size_t>(f.gcount()) != packed.size() ||
        detail::lzfDecompress(packed.data(), packed.size(), raw.data(), stored) != stored) {
      PCL_ERROR("[pcl::PCDReader::read] Size of decompressed lzf data does not match value stored in PCD header\n");
      return -1;
    }
    for (size_t i = 0; i < h.points; ++i) {
      PointT &p = cloud.points[i];
      std::memcpy(&p.x, raw.data() + off_x + 4 * i, 4);
      std::memcpy(&p.y, raw.data() + off_y + 4 * i, 4);
      std::memcpy(&p.z, raw.data() + off_z + 4 * i, 4);
    }
  } else {
    PCL_ERROR("[pcl::PCDReader::read] %s: DATA %s is not supported.\n", file_name.c_str(), h.data.c_str());
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

/** \brief pcl::io::savePCDFileBinaryCompressed (io/src/pcd_io.cpp writeBinaryCompressed): field-major image, LZF */
template <typename PointT>
int savePCDFileBinaryCompressed(const std::string &file_name, const pcl::PointCloud<PointT> &cloud) {
  if (cloud.empty()) {
    PCL_ERROR("[pcl::PCDWriter::writeBinaryCompressed] Input point cloud has no data!\n");
    return -1;
  }
  const size_t n = cloud.points.size();
  if (n * 12 > 0xffffffffull) {
    PCL_ERROR("[pcl::PCDWriter::writeBinaryCompressed] cloud too large for the format's 32-bit sizes\n");
    return -1;
  }
  std::ofstream f(file_name.c_str(), std::ios::binary);
  if (!f) {
    PCL_ERROR("[pcl::PCDWriter::writeBinaryCompressed] Could not open file for writing.\n");
    return -1;
  }
  std::vector<float> soa(n * 3);
  for (size_t i = 0; i < n; ++i) {
    soa[i] = cloud.points[i].x;
    soa[n + i] = cloud.points[i].y;
    soa[2 * n + i] = cloud.points[i].z;
  }
  const size_t raw_bytes = soa.size() * sizeof(float);
  std::vector<unsigned char> packed(raw_bytes + raw_bytes / 16 + 64);
  const size_t packed_bytes = detail::lzfCompress(reinterpret_cast<const unsigned char *>(soa.data()), raw_bytes, packed.data());
  const std::string hdr = detail::generateHeader(cloud) + "DATA binary_compressed\n";
  const std::uint32_t sizes[2] = {static_cast<std::uint32_t>(packed_bytes), static_cast<std::uint32_t>(raw_bytes)};
  f.write(hdr.data(), static_cast<std::streamsize>(hdr.size()));
  f.write(reinterpret_cast<const char *>(sizes), 8);
  f.write(reinterpret_cast<const char *>(packed.data()), static_cast<std::streamsize>(packed_bytes));
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
