// test_pcd_io.cpp - host-only check of include/pcl/io/pcd_io.h: the ASCII v.5 fixture loads, binary, LZF-compressed
// and ASCII files written by the facade read back bit-identically, extra fields are skipped, NaN rows survive.
// usage: test_pcd_io <tests/golden dir> <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include <pcl/io/pcd_io.h>
#include <pcl/point_types.h>

static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { ++g_fail; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

static bool same(const pcl::PointCloud<pcl::PointXYZ> &a, const pcl::PointCloud<pcl::PointXYZ> &b) {
  if (a.size() != b.size() || a.width != b.width || a.height != b.height) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (std::memcmp(a.points[i].data, b.points[i].data, 12) != 0) return false;
  return true;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1], tmp = argv[2];
  pcl::PointCloud<pcl::PointXYZ> bun, back;
  EXPECT(pcl::io::loadPCDFile(dir + "/bun0.pcd", bun) == 0);
  EXPECT(bun.size() == 397 && bun.width == 397 && bun.height == 1 && bun.is_dense);
  EXPECT(bun.points[0].x == 0.0054216f && bun.points[0].y == 0.11349f && bun.points[0].z == 0.040749f);
  EXPECT(bun.points[396].data[3] == 1.f);
  EXPECT(pcl::io::savePCDFileBinary(tmp + "/b.pcd", bun) == 0);
  EXPECT(pcl::io::loadPCDFile(tmp + "/b.pcd", back) == 0 && same(bun, back));
  EXPECT(pcl::io::savePCDFile(tmp + "/a.pcd", bun) == 0);  // ASCII, 8 significant digits: exact for these inputs
  EXPECT(pcl::io::loadPCDFile(tmp + "/a.pcd", back) == 0 && same(bun, back));
  // the written header is the reference's v0.7 layout (impl/pcd_io.hpp:72-121)
  {
    std::ifstream f(tmp + "/b.pcd", std::ios::binary);
    std::string l1, l2, l3;
    std::getline(f, l1); std::getline(f, l2); std::getline(f, l3);
    EXPECT(l1 == "# .PCD v0.7 - Point Cloud Data file format" && l2 == "VERSION 0.7" && l3 == "FIELDS x y z");
  }
  // extra fields and a different order are skipped / mapped by name; NaN rows make the cloud non-dense
  {
    std::ofstream f(tmp + "/c.pcd");
    f << "VERSION .7\nFIELDS intensity z y x rgb\nSIZE 4 4 4 4 4\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 2\n"
         "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 4\nDATA ascii\n9 3 2 1 255\n9 6 5 4 0\n9 nan nan nan 0\n9 -1.5 2.5e-3 1e3 7\n";
  }
  pcl::PointCloud<pcl::PointXYZ> c;
  EXPECT(pcl::io::loadPCDFile(tmp + "/c.pcd", c) == 0);
  EXPECT(c.size() == 4 && c.width == 2 && c.height == 2 && !c.is_dense);
  EXPECT(c.points[0].x == 1.f && c.points[0].y == 2.f && c.points[0].z == 3.f);
  EXPECT(std::isnan(c.points[2].x) && c.points[3].x == 1e3f && c.points[3].y == 2.5e-3f && c.points[3].z == -1.5f);
  EXPECT(pcl::io::savePCDFileBinary(tmp + "/d.pcd", c) == 0 && pcl::io::loadPCDFile(tmp + "/d.pcd", back) == 0);
  EXPECT(same(c, back) && !back.is_dense);
  // DATA binary_compressed (io/src/pcd_io.cpp:1116-1193): LZF byte format, field-major image
  {
    // the byte format itself: a literal run "abc", then a back reference of length 6 at distance 3
    const unsigned char lz[] = {2, 'a', 'b', 'c', (4u << 5) | 0u, 2};
    unsigned char out[16] = {0};
    EXPECT(pcl::io::detail::lzfDecompress(lz, sizeof(lz), out, sizeof(out)) == 9 && std::memcmp(out, "abcabcabc", 9) == 0);
    // a long reference (length code 7 + extension byte): 1 literal 'x', then 40 copies at distance 1
    const unsigned char lz2[] = {0, 'x', (7u << 5) | 0u, 40 - 2 - 7, 0};
    unsigned char out2[64];
    EXPECT(pcl::io::detail::lzfDecompress(lz2, sizeof(lz2), out2, sizeof(out2)) == 41 && out2[40] == 'x');
    EXPECT(pcl::io::detail::lzfDecompress(lz2, sizeof(lz2), out2, 20) == 0);      // output too small
    const unsigned char bad_ref[] = {0, 'x', (1u << 5) | 0u, 5};                  // distance 6 with one byte produced
    EXPECT(pcl::io::detail::lzfDecompress(bad_ref, sizeof(bad_ref), out2, sizeof(out2)) == 0);
    EXPECT(pcl::io::detail::lzfDecompress(lz, 3, out, sizeof(out)) == 0);         // literal run cut short
    // compressor -> decompressor on data with long repeats and on incompressible data
    std::vector<unsigned char> src(100000), packed(100000 + 100000 / 16 + 64), rt(100000);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)((i / 700) * 31 + (i % 7));
    size_t pn = pcl::io::detail::lzfCompress(src.data(), src.size(), packed.data());
    EXPECT(pn > 0 && pn < src.size() / 4);
    EXPECT(pcl::io::detail::lzfDecompress(packed.data(), pn, rt.data(), rt.size()) == src.size() && rt == src);
    unsigned int x = 12345;
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)((x = x * 1664525u + 1013904223u) >> 24);
    pn = pcl::io::detail::lzfCompress(src.data(), src.size(), packed.data());
    EXPECT(pn > 0 && pn <= packed.size());
    EXPECT(pcl::io::detail::lzfDecompress(packed.data(), pn, rt.data(), rt.size()) == src.size() && rt == src);
  }
  EXPECT(pcl::io::savePCDFileBinaryCompressed(tmp + "/z.pcd", bun) == 0);
  EXPECT(pcl::io::loadPCDFile(tmp + "/z.pcd", back) == 0 && same(bun, back) && back.is_dense);
  EXPECT(pcl::io::savePCDFileBinaryCompressed(tmp + "/z2.pcd", c) == 0);
  EXPECT(pcl::io::loadPCDFile(tmp + "/z2.pcd", back) == 0 && same(c, back) && !back.is_dense);
  {
    // a file as PCDWriter lays it out for a padded point type: "_" fields are in the header but not in the image
    const float xs[2] = {1.f, 4.f}, ys[2] = {2.f, 5.f}, zs[2] = {3.f, 6.f}, in[2] = {9.f, 8.f};
    unsigned char image[32], lit[40];
    std::memcpy(image, xs, 8); std::memcpy(image + 8, ys, 8); std::memcpy(image + 16, zs, 8); std::memcpy(image + 24, in, 8);
    lit[0] = 31;  // one literal run of 32 bytes
    std::memcpy(lit + 1, image, 32);
    const unsigned int sizes[2] = {33, 32};
    std::ofstream f(tmp + "/z3.pcd", std::ios::binary);
    f << "VERSION .7\nFIELDS x y z _ intensity _\nSIZE 4 4 4 1 4 1\nTYPE F F F U F U\nCOUNT 1 1 1 4 1 12\nWIDTH 2\nHEIGHT 1\n"
         "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA binary_compressed\n";
    f.write(reinterpret_cast<const char *>(sizes), 8);
    f.write(reinterpret_cast<const char *>(lit), 33);
  }
  EXPECT(pcl::io::loadPCDFile(tmp + "/z3.pcd", back) == 0 && back.size() == 2);
  EXPECT(back.points[0].x == 1.f && back.points[0].y == 2.f && back.points[0].z == 3.f && back.points[1].z == 6.f);
  {
    // truncated / inconsistent compressed payloads are refused
    std::ifstream f(tmp + "/z.pcd", std::ios::binary);
    std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    pcl::console::setVerbosityLevel(pcl::console::L_ALWAYS);
    { std::ofstream g(tmp + "/z4.pcd", std::ios::binary); g << all.substr(0, all.size() - 40); }
    EXPECT(pcl::io::loadPCDFile(tmp + "/z4.pcd", back) == -1);
    std::string wrong = all;
    const size_t d = wrong.find("DATA binary_compressed\n") + 23;
    wrong[d + 4] ^= 0x10;  // the stored uncompressed size no longer matches the header
    { std::ofstream g(tmp + "/z5.pcd", std::ios::binary); g << wrong; }
    EXPECT(pcl::io::loadPCDFile(tmp + "/z5.pcd", back) == -1);
    pcl::console::setVerbosityLevel(pcl::console::L_INFO);
  }
  // errors: missing file, unsupported encoding, empty cloud
  pcl::console::setVerbosityLevel(pcl::console::L_ALWAYS);
  EXPECT(pcl::io::loadPCDFile(tmp + "/missing.pcd", back) == -1);
  {
    std::ofstream f(tmp + "/e.pcd");
    f << "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary_compressed\n";
  }
  EXPECT(pcl::io::loadPCDFile(tmp + "/e.pcd", back) == -1);
  EXPECT(pcl::io::savePCDFile(tmp + "/f.pcd", pcl::PointCloud<pcl::PointXYZ>()) == -1);
  // malformed headers are refused before anything is read or allocated: negative / odd / missing SIZE entries
  // (record offsets would wrap), COUNT 0, a point count the payload cannot hold, negative counts
  {
    const char *bad[] = {
        "VERSION .7\nFIELDS x y z w\nSIZE 4 4 4 -4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 3\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F Q\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 0\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 18446744073709551615\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 4611686018427387904\nDATA binary\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1000000000000\nDATA ascii\n1 2 3\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH -1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n",
        "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS -1\nDATA ascii\n1 2 3\n",
    };
    for (size_t b = 0; b < sizeof(bad) / sizeof(bad[0]); ++b) {
      {
        std::ofstream f(tmp + "/g.pcd", std::ios::binary);
        f << bad[b] << std::string(12, '\0');  // one record of payload
      }
      EXPECT(pcl::io::loadPCDFile(tmp + "/g.pcd", back) == -1);
    }
  }
  std::printf(g_fail ? "test_pcd_io: %d FAILED\n" : "test_pcd_io: all checks passed\n", g_fail);
  return g_fail ? 1 : 0;
}
