// tree_build.cpp - host construction of the left-balanced bucket kd-tree.
//
// Replaces the FLANN KDTreeSingleIndex build behind pcl::KdTreeFLANN::setInputCloud
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:72-74).  The tree SHAPE is ours (exact
// search results do not depend on it): M = ceil(n / L) buckets arranged as a complete binary
// tree in heap order, so the traversal kernel needs no child pointers and both children's
// bounding boxes sit in one 64-byte line.  Every node stores the TIGHT box of its points,
// which prunes far better than split planes for the wave-cooperative traversal.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <numeric>

#include "pclmi_internal.h"

namespace pclmi {
namespace {

struct Builder {
  const float *pts;
  const int32_t *ext;
  size_t n;
  int M;
  std::vector<int32_t> perm;
  HostTree *out;

  // number of leaf ids (M .. 2M-1) inside the subtree rooted at heap id `node`
  int leaves_under(int64_t node) const {
    int64_t lo = node, hi = node, cnt = 0;
    const int64_t first = M, last = 2 * (int64_t)M - 1;
    while (lo <= last) {
      int64_t a = std::max(lo, first), b = std::min(hi, last);
      if (a <= b) cnt += b - a + 1;
      lo = 2 * lo;
      hi = 2 * hi + 1;
    }
    return (int)cnt;
  }

  void set_box(int node, const float lo[3], const float hi[3]) {
    float *b = &out->box[(size_t)node * 8];
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2];
    b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
    b[6] = 0.f; b[7] = 0.f;
  }

  void range_box(size_t a, size_t b, float lo[3], float hi[3]) const {
    for (int d = 0; d < 3; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
    for (size_t i = a; i < b; ++i) {
      const float *p = pts + (size_t)perm[i] * 3;
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], p[d]);
        hi[d] = std::max(hi[d], p[d]);
      }
    }
  }

  // builds the subtree of `node` over perm[a, b); returns its box in lo/hi
  void build(int node, size_t a, size_t b, int par_depth, float lo[3], float hi[3]) {
    if (node >= M) {  // leaf bucket
      const int L = kLeafSize;
      float *rec = &out->leaf[(size_t)(node - M) * 4 * L];
      // deterministic in-bucket order: ascending external index
      std::sort(perm.begin() + a, perm.begin() + b,
                [&](int32_t x, int32_t y) { return ext[x] < ext[y]; });
      size_t cnt = b - a;
      for (int j = 0; j < L; ++j) {
        // group g = j / 4 is one 64-byte line: x[4] y[4] z[4] idx[4]
        float *gp = rec + (j >> 2) * 16 + (j & 3);
        if ((size_t)j < cnt) {
          const float *p = pts + (size_t)perm[a + j] * 3;
          gp[0] = p[0]; gp[4] = p[1]; gp[8] = p[2];
          int32_t e = ext[perm[a + j]];
          std::memcpy(&gp[12], &e, 4);
        } else {
          gp[0] = gp[4] = gp[8] = kPadCoord;
          int32_t e = -1;
          std::memcpy(&gp[12], &e, 4);
        }
      }
      range_box(a, b, lo, hi);
      set_box(node, lo, hi);
      return;
    }
    const int nl = leaves_under(2 * (int64_t)node), nr = leaves_under(2 * (int64_t)node + 1);
    const size_t cnt = b - a;
    // proportional split, every bucket keeps <= L points and (n >= M) at least one
    size_t left = (size_t)std::llround((double)cnt * nl / (double)(nl + nr));
    const size_t max_left = std::min(cnt - (size_t)nr, (size_t)nl * kLeafSize);
    const size_t min_left = std::max((size_t)nl, cnt > (size_t)nr * kLeafSize ? cnt - (size_t)nr * kLeafSize : 0);
    left = std::min(std::max(left, min_left), max_left);
    // split along the widest extent of the tight box, ordered by (coordinate, external index)
    float blo[3], bhi[3];
    range_box(a, b, blo, bhi);
    int dim = 0;
    float ext_best = bhi[0] - blo[0];
    for (int d = 1; d < 3; ++d)
      if (bhi[d] - blo[d] > ext_best) { ext_best = bhi[d] - blo[d]; dim = d; }
    std::nth_element(perm.begin() + a, perm.begin() + a + left, perm.begin() + b,
                     [&](int32_t x, int32_t y) {
                       float cx = pts[(size_t)x * 3 + dim], cy = pts[(size_t)y * 3 + dim];
                       return cx < cy || (cx == cy && ext[x] < ext[y]);
                     });
    float llo[3], lhi[3], rlo[3], rhi[3];
    if (par_depth > 0 && cnt > 65536) {
      auto fut = std::async(std::launch::async,
                            [&] { build(2 * node, a, a + left, par_depth - 1, llo, lhi); });
      build(2 * node + 1, a + left, b, par_depth - 1, rlo, rhi);
      fut.get();
    } else {
      build(2 * node, a, a + left, 0, llo, lhi);
      build(2 * node + 1, a + left, b, 0, rlo, rhi);
    }
    for (int d = 0; d < 3; ++d) {
      lo[d] = std::min(llo[d], rlo[d]);
      hi[d] = std::max(lhi[d], rhi[d]);
    }
    set_box(node, lo, hi);
  }
};

}  // namespace

void build_host_tree(const float *pts, const int32_t *ext, size_t n, HostTree *out) {
  out->M = 0;
  out->box.clear();
  out->leaf.clear();
  for (int d = 0; d < 3; ++d) out->bb_lo[d] = out->bb_hi[d] = 0.f;
  if (n == 0) return;
  Builder b;
  b.pts = pts;
  b.ext = ext;
  b.n = n;
  b.M = (int)((n + kLeafSize - 1) / kLeafSize);
  b.out = out;
  b.perm.resize(n);
  std::iota(b.perm.begin(), b.perm.end(), 0);
  out->M = b.M;
  out->box.assign((size_t)2 * b.M * 8, 0.f);
  out->leaf.assign((size_t)b.M * 4 * kLeafSize, 0.f);
  float lo[3], hi[3];
  b.build(1, 0, n, /*par_depth=*/3, lo, hi);
  // id 0 is unused; give it the root box so that a stray read is harmless
  std::memcpy(&out->box[0], &out->box[8], 8 * sizeof(float));
  for (int d = 0; d < 3; ++d) { out->bb_lo[d] = lo[d]; out->bb_hi[d] = hi[d]; }
}

}  // namespace pclmi
