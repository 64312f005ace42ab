// pclmi_internal.h - shared declarations of libpclmi.so (not part of the public ABI).
#ifndef PCLMI_INTERNAL_H_
#define PCLMI_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <vector>

#include "pclmi.h"

namespace pclmi {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PCLMI_HIP(call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) return ::pclmi::hip_fail(e_, #call, __FILE__, __LINE__);       \
  } while (0)

#define PCLMI_CHECK(cond, code, ...)                                                     \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      ::pclmi::set_error(__VA_ARGS__);                                                   \
      return (code);                                                                     \
    }                                                                                    \
  } while (0)

// ---- device tree ------------------------------------------------------------------------
// Left-balanced bucket kd-tree in implicit (heap) order: node ids 1 .. 2M-1, children of i
// are 2i and 2i+1, ids >= M are leaves (leaf number = id - M).  See DESIGN.md "Data layout".
constexpr int kLeafSize = 16;        // points per leaf bucket (L)
constexpr float kPadCoord = 3.0e38f; // coordinate of unused bucket slots: d2 overflows to +inf

struct TreeView {
  const float *box;  // [2M][8]  lo.xyz, hi.xyz, 0, 0   (tight bounding box of the subtree)
  const float *leaf; // [M][L/4][16]: per group of 4 points x[4], y[4], z[4], idx[4] (int bits;
                     // external row index, -1 in padded slots) - one 64-byte line per group
  int M;             // number of leaves (>= 1 when n > 0)
  int n;             // number of indexed points
};

struct Workspace; // per-call scratch (Morton keys, permutations, sort temp), pooled per tree

}  // namespace pclmi

struct pclmi_tree {
  int device = 0;
  size_t n = 0;          // indexed points
  size_t rows = 0;       // rows of the original cloud
  int M = 0;             // leaves
  float *d_box = nullptr;
  float *d_leaf = nullptr;
  float *d_cloud = nullptr;  // [rows][4] copy of the original cloud (x, y, z, 1) for ICP gathers
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  // workspace pool (const search calls are re-entrant: each takes a workspace)
  mutable std::mutex mu;
  mutable std::vector<pclmi::Workspace *> pool;

  pclmi::TreeView view() const { return pclmi::TreeView{d_box, d_leaf, M, (int)n}; }
};

namespace pclmi {

struct Workspace {
  // query-ordering scratch (counting sort by Morton cell)
  uint32_t *keys_in = nullptr;  // cell id per query
  int32_t *vals_out = nullptr;  // sorted slot -> query row
  uint32_t *hist = nullptr;     // per-cell counts / running offsets
  void *sort_tmp = nullptr;     // hipcub scan temp storage
  size_t cap_q = 0, cap_bins = 0, cap_tmp = 0;
  // generic scratch (large-k heaps, radius counts, ...)
  void *scratch = nullptr;
  size_t cap_scratch = 0;
  int ensure_queries(size_t nq, size_t nbins);
  int ensure_scratch(size_t bytes);
  void release();
};

Workspace *acquire_ws(const pclmi_tree *t);
void release_ws(const pclmi_tree *t, Workspace *w);
struct WsGuard {
  const pclmi_tree *t;
  Workspace *w;
  explicit WsGuard(const pclmi_tree *tt) : t(tt), w(acquire_ws(tt)) {}
  ~WsGuard() { release_ws(t, w); }
};

// ---- host tree builder (tree_build.cpp) --------------------------------------------------
struct HostTree {
  int M = 0;
  std::vector<float> box;   // 2M * 8
  std::vector<float> leaf;  // M * 4 * L
  float bb_lo[3], bb_hi[3];
};
// pts: n packed xyz (finite), ext: n external row indices
void build_host_tree(const float *pts, const int32_t *ext, size_t n, HostTree *out);

// ---- kernels launchers (knn_kernels.hip) -------------------------------------------------
// order: nullable device permutation (sorted slot -> query row)
int launch_morton_order(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq,
                        size_t q_stride_f, hipStream_t s, const int32_t **d_order);
int launch_knn(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_f,
               const int32_t *d_order, int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s);

}  // namespace pclmi
#endif
