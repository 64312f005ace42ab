/*
 * pcl_oracle.c - CPU restatement of the reference kNN / radius / ICP path.
 * TEST INFRASTRUCTURE ONLY - see pcl_oracle.h for the contract and the pinning list.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference tree).  Third-party arithmetic: the tree build / traversal / result sets
 * live in FLANN (>= 1.7.0 required by CMakeLists.txt:208; 3rdparty/flann/Makefile pins
 * 1.6.11; its sources are NOT in the reference tree).  The kd-tree below restates the
 * published KDTreeSingleIndex algorithm (middle split on the widest bbox dimension, leaf
 * <= leaf_max, incremental per-dimension lower bounds) and is anchored on the reference's
 * own call sites (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:72-74,97-99,137-141) and
 * golden vectors; correctness is defined by orc_knn_brute / orc_radius_brute, the
 * restatement of search/include/pcl/search/impl/brute_force.hpp.
 */
#include "pcl_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* distance: flann::L2_Simple<float> - sequential sum of squared differences in fp32     */
static inline float sqdist3(const float *a, const float *b) {
  float dx = a[0] - b[0];
  float dy = a[1] - b[1];
  float dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

/* total order on (d2, index) */
static inline int key_less(float da, int32_t ia, float db, int32_t ib) {
  return da < db || (da == db && ia < ib);
}

void orc_free(void *p) { free(p); }

/* k-best list, ascending by (d2, idx) */
typedef struct {
  int k, n;
  float *d;
  int32_t *i;
} kbest;

static inline void kbest_push(kbest *r, float d, int32_t i) {
  int pos;
  if (r->n == r->k) {
    if (!key_less(d, i, r->d[r->k - 1], r->i[r->k - 1])) return;
    pos = r->k - 1;
  } else {
    pos = r->n++;
  }
  while (pos > 0 && key_less(d, i, r->d[pos - 1], r->i[pos - 1])) {
    r->d[pos] = r->d[pos - 1];
    r->i[pos] = r->i[pos - 1];
    --pos;
  }
  r->d[pos] = d;
  r->i[pos] = i;
}

int orc_knn_brute(const float *pts, int64_t n, int ps, const float *q, int64_t nq, int qs,
                  int k, int32_t *idx, float *d2) {
  if (k > n) k = (int)n; /* impl/kdtree_flann.hpp:85-86 */
  if (k <= 0) return 0;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t j = 0; j < nq; ++j) {
    kbest r = {k, 0, d2 + j * k, idx + j * k};
    const float *qq = q + j * qs;
    for (int64_t i = 0; i < n; ++i) kbest_push(&r, sqdist3(qq, pts + i * ps), (int32_t)i);
  }
  return k;
}

/* growable (d2, idx) list for radius results */
typedef struct {
  int64_t n, cap;
  float *d;
  int32_t *i;
} rlist;

static void rlist_push(rlist *r, float d, int32_t i) {
  if (r->n == r->cap) {
    r->cap = r->cap ? r->cap * 2 : 32;
    r->d = (float *)realloc(r->d, (size_t)r->cap * sizeof(float));
    r->i = (int32_t *)realloc(r->i, (size_t)r->cap * sizeof(int32_t));
  }
  r->d[r->n] = d;
  r->i[r->n] = i;
  r->n++;
}

typedef struct {
  float d;
  int32_t i;
} di_pair;

static int cmp_di(const void *a, const void *b) {
  const di_pair *x = (const di_pair *)a, *y = (const di_pair *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}
static int cmp_i(const void *a, const void *b) {
  const di_pair *x = (const di_pair *)a, *y = (const di_pair *)b;
  return (x->i > y->i) - (x->i < y->i);
}

/* order / truncate one query's radius result the way the reference adapter exposes it:
 * max_nn>0 keeps the nearest max_nn (impl/kdtree_flann.hpp:131-135), sorted_ chooses
 * ascending distance (kdtree_flann.h:85,138-144); unsorted output is ascending index. */
static void rlist_finish(rlist *r, uint32_t max_nn, int sorted) {
  if (r->n == 0) return;
  di_pair *t = (di_pair *)malloc((size_t)r->n * sizeof(di_pair));
  for (int64_t a = 0; a < r->n; ++a) {
    t[a].d = r->d[a];
    t[a].i = r->i[a];
  }
  int truncated = max_nn > 0 && (int64_t)max_nn < r->n;
  if (sorted || truncated) qsort(t, (size_t)r->n, sizeof(di_pair), cmp_di);
  if (truncated) r->n = max_nn;
  if (!sorted) qsort(t, (size_t)r->n, sizeof(di_pair), cmp_i);
  for (int64_t a = 0; a < r->n; ++a) {
    r->d[a] = t[a].d;
    r->i[a] = t[a].i;
  }
  free(t);
}

static int gather_rlists(rlist *lists, int64_t nq, int64_t *offsets, int32_t **idx, float **d2) {
  offsets[0] = 0;
  for (int64_t j = 0; j < nq; ++j) offsets[j + 1] = offsets[j] + lists[j].n;
  int64_t total = offsets[nq];
  *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
  *d2 = (float *)malloc((size_t)(total ? total : 1) * sizeof(float));
  for (int64_t j = 0; j < nq; ++j) {
    if (lists[j].n) {
      memcpy(*idx + offsets[j], lists[j].i, (size_t)lists[j].n * sizeof(int32_t));
      memcpy(*d2 + offsets[j], lists[j].d, (size_t)lists[j].n * sizeof(float));
    }
    free(lists[j].d);
    free(lists[j].i);
  }
  return 0;
}

int orc_radius_brute(const float *pts, int64_t n, int ps, const float *q, int64_t nq, int qs,
                     float r2, uint32_t max_nn, int sorted, int64_t *offsets, int32_t **idx,
                     float **d2) {
  if (max_nn == 0 || (int64_t)max_nn > n) max_nn = 0; /* impl/kdtree_flann.hpp:125-133 */
  rlist *lists = (rlist *)calloc((size_t)(nq ? nq : 1), sizeof(rlist));
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t j = 0; j < nq; ++j) {
    const float *qq = q + j * qs;
    for (int64_t i = 0; i < n; ++i) {
      float d = sqdist3(qq, pts + i * ps);
      if (d < r2) rlist_push(&lists[j], d, (int32_t)i); /* strict: FLANN radius rule */
    }
    rlist_finish(&lists[j], max_nn, sorted);
  }
  int rc = gather_rlists(lists, nq, offsets, idx, d2);
  free(lists);
  return rc;
}

/* ------------------------------------------------------------------------------------ */
/* FLANN-style KDTreeSingleIndex (see file header)                                        */
typedef struct {
  int32_t left, right;   /* leaf: point range [left,right) in vind; inner: unused */
  int32_t child1, child2; /* -1 for leaves */
  int32_t divfeat;
  float divlow, divhigh;
} knode;

struct orc_tree {
  int64_t n;
  int leaf_max;
  float *data; /* n x 3 packed, REORDERED so leaves are contiguous */
  int32_t *vind; /* vind[pos] = original index */
  knode *nodes;
  int64_t nnodes, capnodes;
  float bb_lo[3], bb_hi[3];
};

typedef struct {
  float lo[3], hi[3];
} bbox3;

static int32_t new_node(orc_tree *t) {
  if (t->nnodes == t->capnodes) {
    t->capnodes = t->capnodes ? t->capnodes * 2 : 1024;
    t->nodes = (knode *)realloc(t->nodes, (size_t)t->capnodes * sizeof(knode));
  }
  return (int32_t)t->nnodes++;
}

static void plane_split(const float *pts, int32_t *ind, int count, int feat, float cut,
                        int *lim1, int *lim2) {
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && pts[(int64_t)ind[left] * 3 + feat] < cut) ++left;
    while (left <= right && pts[(int64_t)ind[right] * 3 + feat] >= cut) --right;
    if (left > right) break;
    int32_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && pts[(int64_t)ind[left] * 3 + feat] <= cut) ++left;
    while (left <= right && pts[(int64_t)ind[right] * 3 + feat] > cut) --right;
    if (left > right) break;
    int32_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim2 = left;
}

static void middle_split(const float *pts, int32_t *ind, int count, const bbox3 *bb, int *index,
                         int *cutfeat, float *cutval) {
  const float EPS = 0.00001f;
  float max_span = bb->hi[0] - bb->lo[0];
  for (int d = 1; d < 3; ++d) {
    float s = bb->hi[d] - bb->lo[d];
    if (s > max_span) max_span = s;
  }
  float max_spread = -1.0f;
  *cutfeat = 0;
  for (int d = 0; d < 3; ++d) {
    float s = bb->hi[d] - bb->lo[d];
    if (s > (1.0f - EPS) * max_span) {
      float mn = pts[(int64_t)ind[0] * 3 + d], mx = mn;
      for (int a = 1; a < count; ++a) {
        float v = pts[(int64_t)ind[a] * 3 + d];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
      }
      if (mx - mn > max_spread) {
        *cutfeat = d;
        max_spread = mx - mn;
      }
    }
  }
  int f = *cutfeat;
  float split = (bb->lo[f] + bb->hi[f]) / 2;
  float mn = pts[(int64_t)ind[0] * 3 + f], mx = mn;
  for (int a = 1; a < count; ++a) {
    float v = pts[(int64_t)ind[a] * 3 + f];
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
  if (split < mn) split = mn;
  else if (split > mx) split = mx;
  *cutval = split;
  int lim1, lim2;
  plane_split(pts, ind, count, f, split, &lim1, &lim2);
  if (lim1 > count / 2) *index = lim1;
  else if (lim2 < count / 2) *index = lim2;
  else *index = count / 2;
}

static int32_t divide_tree(orc_tree *t, const float *pts, int32_t *ind, int left, int right,
                           bbox3 *bb) {
  int32_t id = new_node(t);
  if (right - left <= t->leaf_max) {
    knode nd = {left, right, -1, -1, 0, 0.f, 0.f};
    for (int d = 0; d < 3; ++d) {
      float mn = pts[(int64_t)ind[left] * 3 + d], mx = mn;
      for (int a = left + 1; a < right; ++a) {
        float v = pts[(int64_t)ind[a] * 3 + d];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
      }
      bb->lo[d] = mn;
      bb->hi[d] = mx;
    }
    t->nodes[id] = nd;
    return id;
  }
  int idx, cutfeat;
  float cutval;
  middle_split(pts, ind + left, right - left, bb, &idx, &cutfeat, &cutval);
  bbox3 lb = *bb, rb = *bb;
  lb.hi[cutfeat] = cutval;
  int32_t c1 = divide_tree(t, pts, ind, left, left + idx, &lb);
  rb.lo[cutfeat] = cutval;
  int32_t c2 = divide_tree(t, pts, ind, left + idx, right, &rb);
  knode nd = {0, 0, c1, c2, cutfeat, lb.hi[cutfeat], rb.lo[cutfeat]};
  t->nodes[id] = nd;
  for (int d = 0; d < 3; ++d) {
    bb->lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
    bb->hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
  }
  return id;
}

orc_tree *orc_tree_build(const float *pts, int64_t n, int ps, int leaf_max) {
  orc_tree *t = (orc_tree *)calloc(1, sizeof(orc_tree));
  t->n = n;
  t->leaf_max = leaf_max > 0 ? leaf_max : 15; /* impl/kdtree_flann.hpp:72-74 */
  if (n <= 0) return t;
  float *packed = (float *)malloc((size_t)n * 3 * sizeof(float));
  t->vind = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  for (int64_t i = 0; i < n; ++i) {
    packed[i * 3 + 0] = pts[i * ps + 0];
    packed[i * 3 + 1] = pts[i * ps + 1];
    packed[i * 3 + 2] = pts[i * ps + 2];
    t->vind[i] = (int32_t)i;
  }
  bbox3 bb;
  for (int d = 0; d < 3; ++d) {
    float mn = packed[d], mx = mn;
    for (int64_t i = 1; i < n; ++i) {
      float v = packed[i * 3 + d];
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
    bb.lo[d] = mn;
    bb.hi[d] = mx;
  }
  divide_tree(t, packed, t->vind, 0, (int)n, &bb);
  for (int d = 0; d < 3; ++d) {
    t->bb_lo[d] = bb.lo[d];
    t->bb_hi[d] = bb.hi[d];
  }
  /* reorder the data so every leaf scans contiguous memory (FLANN's reorder_ option) */
  t->data = (float *)malloc((size_t)n * 3 * sizeof(float));
  for (int64_t i = 0; i < n; ++i) memcpy(t->data + i * 3, packed + (int64_t)t->vind[i] * 3, 12);
  free(packed);
  return t;
}

void orc_tree_free(orc_tree *t) {
  if (!t) return;
  free(t->data);
  free(t->vind);
  free(t->nodes);
  free(t);
}

static float initial_dists(const orc_tree *t, const float *q, float dists[3]) {
  float sum = 0.f;
  for (int d = 0; d < 3; ++d) {
    dists[d] = 0.f;
    if (q[d] < t->bb_lo[d]) dists[d] = (q[d] - t->bb_lo[d]) * (q[d] - t->bb_lo[d]);
    if (q[d] > t->bb_hi[d]) dists[d] = (q[d] - t->bb_hi[d]) * (q[d] - t->bb_hi[d]);
    sum += dists[d];
  }
  return sum;
}

static void knn_level(const orc_tree *t, const float *q, int32_t node, float mindist,
                      float dists[3], kbest *r) {
  const knode *nd = &t->nodes[node];
  if (nd->child1 < 0) {
    for (int a = nd->left; a < nd->right; ++a)
      kbest_push(r, sqdist3(q, t->data + (int64_t)a * 3), t->vind[a]);
    return;
  }
  int f = nd->divfeat;
  float val = q[f];
  float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
  int32_t best, other;
  float cut;
  if (diff1 + diff2 < 0) {
    best = nd->child1;
    other = nd->child2;
    cut = diff2 * diff2;
  } else {
    best = nd->child2;
    other = nd->child1;
    cut = diff1 * diff1;
  }
  knn_level(t, q, best, mindist, dists, r);
  float dst = dists[f];
  mindist = mindist + cut - dst;
  dists[f] = cut;
  /* non-strict so that an equal-distance, lower-index candidate is still reached; and with a few ulps of slack:
   * FLANN's incremental `mindist + cut - dst` is rounded three times and can come out an ulp ABOVE the distance of a
   * point that lies on the cut (duplicated points on either side of a plane: found by tools/stress_search.py), which
   * would prune a tie the (d2, index) rule has to see.  The leaf tests are exact, so the slack only costs visits. */
  if (r->n < r->k || mindist * 0.999996f <= r->d[r->k - 1]) knn_level(t, q, other, mindist, dists, r);
  dists[f] = dst;
}

static void radius_level(const orc_tree *t, const float *q, int32_t node, float mindist,
                         float dists[3], float r2, rlist *r) {
  const knode *nd = &t->nodes[node];
  if (nd->child1 < 0) {
    for (int a = nd->left; a < nd->right; ++a) {
      float d = sqdist3(q, t->data + (int64_t)a * 3);
      if (d < r2) rlist_push(r, d, t->vind[a]);
    }
    return;
  }
  int f = nd->divfeat;
  float val = q[f];
  float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
  int32_t best, other;
  float cut;
  if (diff1 + diff2 < 0) {
    best = nd->child1;
    other = nd->child2;
    cut = diff2 * diff2;
  } else {
    best = nd->child2;
    other = nd->child1;
    cut = diff1 * diff1;
  }
  radius_level(t, q, best, mindist, dists, r2, r);
  float dst = dists[f];
  mindist = mindist + cut - dst;
  dists[f] = cut;
  if (mindist * 0.999996f <= r2) radius_level(t, q, other, mindist, dists, r2, r); /* slack: see knn_level */
  dists[f] = dst;
}

int orc_tree_knn(const orc_tree *t, const float *q, int64_t nq, int qs, int k, int32_t *idx,
                 float *d2, int nthreads) {
  if (k > t->n) k = (int)t->n;
  if (k <= 0) return 0;
  if (nthreads <= 1) {
    for (int64_t j = 0; j < nq; ++j) {
      kbest r = {k, 0, d2 + j * k, idx + j * k};
      float dists[3];
      float md = initial_dists(t, q + j * qs, dists);
      knn_level(t, q + j * qs, 0, md, dists, &r);
    }
  } else {
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int64_t j = 0; j < nq; ++j) {
      kbest r = {k, 0, d2 + j * k, idx + j * k};
      float dists[3];
      float md = initial_dists(t, q + j * qs, dists);
      knn_level(t, q + j * qs, 0, md, dists, &r);
    }
  }
  return k;
}

int orc_tree_radius(const orc_tree *t, const float *q, int64_t nq, int qs, float r2,
                    uint32_t max_nn, int sorted, int64_t *offsets, int32_t **idx, float **d2,
                    int nthreads) {
  if (max_nn == 0 || (int64_t)max_nn > t->n) max_nn = 0;
  rlist *lists = (rlist *)calloc((size_t)(nq ? nq : 1), sizeof(rlist));
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
  for (int64_t j = 0; j < nq; ++j) {
    if (t->n > 0) {
      float dists[3];
      float md = initial_dists(t, q + j * qs, dists);
      radius_level(t, q + j * qs, 0, md, dists, r2, &lists[j]);
    }
    rlist_finish(&lists[j], max_nn, sorted);
  }
  int rc = gather_rlists(lists, nq, offsets, idx, d2);
  free(lists);
  return rc;
}

/* ------------------------------------------------------------------------------------ */
/* transforms.hpp:184-191 - rot * p + trans in fp32; Eigen evaluates the 3x3 product as a
 * sum over columns, ((c0*x + c1*y) + c2*z), then adds the translation.                   */
static inline void xform_point(const float T[16], const float *p, float *o) {
  float x = p[0], y = p[1], z = p[2];
  o[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  o[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  o[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

void orc_transform(float *pts, int64_t n, int stride, const float T[16]) {
  for (int64_t i = 0; i < n; ++i) {
    float o[3];
    xform_point(T, pts + i * stride, o);
    pts[i * stride + 0] = o[0];
    pts[i * stride + 1] = o[1];
    pts[i * stride + 2] = o[2];
  }
}

static void mat4_identity(float T[16]) {
  memset(T, 0, 16 * sizeof(float));
  T[0] = T[5] = T[10] = T[15] = 1.f;
}

static void mat4_mul(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      R[r * 4 + c] = s;
    }
  memcpy(C, R, sizeof(R));
}

/* 3x3 SVD, one-sided Jacobi in fp32 (stands in for Eigen::JacobiSVD<Matrix3f>, a
 * third-party routine not in the reference tree; singular values sorted descending as
 * Eigen does so that "column 2" below is the smallest one).  A = U diag(s) V'.           */
static void svd3(const float A[9], float U[9], float S[3], float V[9]) {
  float a[9];
  /* Eigen::JacobiSVD works on matrix / max|coefficient| (JacobiSVD.h "m_scaledMatrix"): without it the products
   * alpha * beta below overflow fp32 for the H of a few 100k points of magnitude 1e3 and no rotation is applied */
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) scale = fabsf(A[i]) > scale ? fabsf(A[i]) : scale;
  if (!(scale > 0.f)) scale = 1.f;
  for (int i = 0; i < 9; ++i) a[i] = A[i] / scale;
  float v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        float alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += a[r * 3 + p] * a[r * 3 + p];
          beta += a[r * 3 + q] * a[r * 3 + q];
          gamma += a[r * 3 + p] * a[r * 3 + q];
        }
        if (fabsf(gamma) <= 1e-30f || fabsf(gamma) <= FLT_EPSILON * 0.25f * sqrtf(alpha * beta))
          continue;
        off += fabsf(gamma);
        float zeta = (beta - alpha) / (2.f * gamma);
        float tt = (zeta >= 0 ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
        float c = 1.f / sqrtf(1.f + tt * tt), s = c * tt;
        for (int r = 0; r < 3; ++r) {
          float x = a[r * 3 + p], y = a[r * 3 + q];
          a[r * 3 + p] = c * x - s * y;
          a[r * 3 + q] = s * x + c * y;
          x = v[r * 3 + p];
          y = v[r * 3 + q];
          v[r * 3 + p] = c * x - s * y;
          v[r * 3 + q] = s * x + c * y;
        }
      }
    if (off == 0.f) break;
  }
  float s[3];
  for (int c = 0; c < 3; ++c)
    s[c] = sqrtf(a[c] * a[c] + a[3 + c] * a[3 + c] + a[6 + c] * a[6 + c]);
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (s[ord[j]] > s[ord[i]]) {
        int t = ord[i];
        ord[i] = ord[j];
        ord[j] = t;
      }
  for (int c = 0; c < 3; ++c) {
    S[c] = s[ord[c]];
    for (int r = 0; r < 3; ++r) {
      V[r * 3 + c] = v[r * 3 + ord[c]];
      U[r * 3 + c] = S[c] > 0.f ? a[r * 3 + ord[c]] / S[c] : 0.f;
    }
  }
  for (int c = 0; c < 3; ++c) S[c] *= scale; /* (the rank tests below compare singular values with each other) */
  /* rank-deficient input (3-point RANSAC samples give rank 2): complete U to a full
   * orthonormal basis, which is what ComputeFullU returns */
  float tiny = S[0] * 1e-6f;
  if (S[2] <= tiny) {
    if (S[1] <= tiny) { /* rank <= 1: build any orthonormal completion */
      float u0[3] = {U[0], U[3], U[6]};
      if (S[0] <= 0.f) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
      int m = fabsf(u0[0]) < fabsf(u0[1]) ? (fabsf(u0[0]) < fabsf(u0[2]) ? 0 : 2)
                                            : (fabsf(u0[1]) < fabsf(u0[2]) ? 1 : 2);
      float e[3] = {0, 0, 0};
      e[m] = 1.f;
      float w[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2],
                    u0[0] * e[1] - u0[1] * e[0]};
      float nw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      U[1] = w[0] / nw; U[4] = w[1] / nw; U[7] = w[2] / nw;
    }
    float u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    U[2] = u0[1] * u1[2] - u0[2] * u1[1];
    U[5] = u0[2] * u1[0] - u0[0] * u1[2];
    U[8] = u0[0] * u1[1] - u0[1] * u1[0];
  }
}

static float det3(const float M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* impl/transformation_estimation_svd.hpp:137-168 given H and the two centroids */
static void rigid_from_correlation(const float H[9], const float cs[3], const float ct[3],
                                   float T[16]) {
  float U[9], S[3], V[9];
  svd3(H, U, S, V);
  if (det3(U) * det3(V) < 0)
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  float R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += V[r * 3 + k] * U[c * 3 + k]; /* V * U' */
      R[r * 3 + c] = s;
    }
  mat4_identity(T);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
    float rc = (R[r * 3 + 0] * cs[0] + R[r * 3 + 1] * cs[1]) + R[r * 3 + 2] * cs[2];
    T[r * 4 + 3] = ct[r] - rc;
  }
}

/* centroid.hpp:48-75 (fp32 running sums / n), demean :720-748, H = S * T' in fp32 */
void orc_estimate_rigid_svd(const float *src, int ss, const int32_t *sidx, const float *tgt,
                            int ts, const int32_t *tidx, int64_t n, float T[16]) {
  float cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i) {
    const float *s = src + (int64_t)(sidx ? sidx[i] : i) * ss;
    const float *t = tgt + (int64_t)(tidx ? tidx[i] : i) * ts;
    for (int d = 0; d < 3; ++d) {
      cs[d] += s[d];
      ct[d] += t[d];
    }
  }
  for (int d = 0; d < 3; ++d) {
    cs[d] /= (float)n;
    ct[d] /= (float)n;
  }
  float H[9] = {0};
  for (int64_t i = 0; i < n; ++i) {
    const float *s = src + (int64_t)(sidx ? sidx[i] : i) * ss;
    const float *t = tgt + (int64_t)(tidx ? tidx[i] : i) * ts;
    float sd[3] = {s[0] - cs[0], s[1] - cs[1], s[2] - cs[2]};
    float td[3] = {t[0] - ct[0], t[1] - ct[1], t[2] - ct[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[r * 3 + c] += sd[r] * td[c];
  }
  rigid_from_correlation(H, cs, ct, T);
}

/* The same estimator instantiated with Scalar = double (the template parameter of
 * TransformationEstimationSVD, transformation_estimation_svd.h:57): centroids, demeaned clouds,
 * H and the Jacobi SVD all in fp64, result stored as fp32.  Used to pin the HIP path's fp64
 * moment reduction to 1e-5 where the fp32 instantiation's own accumulation error is larger. */
static void svd3d(const double A[9], double U[9], double S[3], double V[9]) {
  double a[9];
  memcpy(a, A, sizeof(a));
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += a[r * 3 + p] * a[r * 3 + p];
          beta += a[r * 3 + q] * a[r * 3 + q];
          gamma += a[r * 3 + p] * a[r * 3 + q];
        }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= DBL_EPSILON * 0.25 * sqrt(alpha * beta)) continue;
        off += fabs(gamma);
        double zeta = (beta - alpha) / (2.0 * gamma);
        double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + tt * tt), sn = c * tt;
        for (int r = 0; r < 3; ++r) {
          double x = a[r * 3 + p], y = a[r * 3 + q];
          a[r * 3 + p] = c * x - sn * y;
          a[r * 3 + q] = sn * x + c * y;
          x = v[r * 3 + p];
          y = v[r * 3 + q];
          v[r * 3 + p] = c * x - sn * y;
          v[r * 3 + q] = sn * x + c * y;
        }
      }
    if (off == 0.0) break;
  }
  double sv[3];
  for (int c = 0; c < 3; ++c) sv[c] = sqrt(a[c] * a[c] + a[3 + c] * a[3 + c] + a[6 + c] * a[6 + c]);
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (sv[ord[j]] > sv[ord[i]]) {
        int t = ord[i];
        ord[i] = ord[j];
        ord[j] = t;
      }
  for (int c = 0; c < 3; ++c) {
    S[c] = sv[ord[c]];
    for (int r = 0; r < 3; ++r) {
      V[r * 3 + c] = v[r * 3 + ord[c]];
      U[r * 3 + c] = S[c] > 0.0 ? a[r * 3 + ord[c]] / S[c] : 0.0;
    }
  }
  if (S[2] <= S[0] * 1e-13) { /* rank 2 (planar / 3-point input): complete U, as ComputeFullU does */
    double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    U[2] = u0[1] * u1[2] - u0[2] * u1[1];
    U[5] = u0[2] * u1[0] - u0[0] * u1[2];
    U[8] = u0[0] * u1[1] - u0[1] * u1[0];
  }
}

void orc_estimate_rigid_svd_f64(const float *src, int ss, const int32_t *sidx, const float *tgt,
                                int ts, const int32_t *tidx, int64_t n, float T[16]) {
  double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i) {
    const float *s = src + (int64_t)(sidx ? sidx[i] : i) * ss;
    const float *t = tgt + (int64_t)(tidx ? tidx[i] : i) * ts;
    for (int d = 0; d < 3; ++d) {
      cs[d] += (double)s[d];
      ct[d] += (double)t[d];
    }
  }
  for (int d = 0; d < 3; ++d) {
    cs[d] /= (double)n;
    ct[d] /= (double)n;
  }
  double H[9] = {0};
  for (int64_t i = 0; i < n; ++i) {
    const float *s = src + (int64_t)(sidx ? sidx[i] : i) * ss;
    const float *t = tgt + (int64_t)(tidx ? tidx[i] : i) * ts;
    double sd[3] = {s[0] - cs[0], s[1] - cs[1], s[2] - cs[2]};
    double td[3] = {t[0] - ct[0], t[1] - ct[1], t[2] - ct[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[r * 3 + c] += sd[r] * td[c];
  }
  double U[9], S[3], V[9];
  svd3d(H, U, S, V);
  double dU = U[0] * (U[4] * U[8] - U[5] * U[7]) - U[1] * (U[3] * U[8] - U[5] * U[6]) +
              U[2] * (U[3] * U[7] - U[4] * U[6]);
  double dV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) +
              V[2] * (V[3] * V[7] - V[4] * V[6]);
  if (dU * dV < 0)
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  mat4_identity(T);
  for (int r = 0; r < 3; ++r) {
    double R[3];
    for (int c = 0; c < 3; ++c) {
      double sum = 0.0;
      for (int k = 0; k < 3; ++k) sum += V[r * 3 + k] * U[c * 3 + k];
      R[c] = sum;
      T[r * 4 + c] = (float)sum;
    }
    T[r * 4 + 3] = (float)(ct[r] - ((R[0] * cs[0] + R[1] * cs[1]) + R[2] * cs[2]));
  }
}

/* ------------------------------------------------------------------------------------ */
int64_t orc_correspondences(const float *src, int64_t ns, int ss, const float *tgt, int64_t nt,
                            int ts, double max_dist, int reciprocal, orc_corr *out) {
  double max2 = max_dist * max_dist;
  orc_tree *tt = orc_tree_build(tgt, nt, ts, 15);
  orc_tree *st = reciprocal ? orc_tree_build(src, ns, ss, 15) : NULL;
  int64_t cnt = 0;
  for (int64_t i = 0; i < ns; ++i) {
    int32_t j;
    float d;
    orc_tree_knn(tt, src + i * ss, 1, ss, 1, &j, &d, 1);
    if (d > max2) continue; /* non-strict keep: correspondence_estimation.hpp:108 */
    if (reciprocal) {
      int32_t jr;
      float dr;
      orc_tree_knn(st, tgt + (int64_t)j * ts, 1, ts, 1, &jr, &dr, 1);
      if (dr > max2 || jr != (int32_t)i) continue; /* :188 */
    }
    out[cnt].index_query = (int32_t)i;
    out[cnt].index_match = j;
    out[cnt].distance = d;
    ++cnt;
  }
  orc_tree_free(tt);
  orc_tree_free(st);
  return cnt;
}

int64_t orc_reject_distance(const orc_corr *in, int64_t n, float max_dist, orc_corr *out) {
  float m2 = max_dist * max_dist; /* correspondence_rejection_distance.h:88 */
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (in[i].distance < m2) out[c++] = in[i]; /* _distance.hpp:62 strict */
  return c;
}

/* impl/correspondence_rejection_surface_normal.hpp:44-66 + correspondence_rejection.h:254-262: keep iff the fp32 dot
 * product of the two normals, widened to double, is > threshold */
int64_t orc_reject_surface_normal(const orc_corr *in, int64_t n, const float *sn, int sns, const float *tn, int tns,
                                  double threshold, orc_corr *out) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float *a = sn + (int64_t)in[i].index_query * sns, *b = tn + (int64_t)in[i].index_match * tns;
    float dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    if ((double)dot > threshold) out[c++] = in[i];
  }
  return c;
}

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int64_t orc_reject_median(const orc_corr *in, int64_t n, double factor, orc_corr *out,
                          double *median) {
  if (n == 0) return 0;
  double *d = (double *)malloc((size_t)n * sizeof(double));
  for (int64_t i = 0; i < n; ++i) d[i] = in[i].distance;
  qsort(d, (size_t)n, sizeof(double), cmp_double); /* nth_element(n/2) value */
  double med = d[n / 2];
  free(d);
  if (median) *median = med;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if ((double)in[i].distance <= med * factor) out[c++] = in[i]; /* _median_distance.hpp:64 */
  return c;
}

static int cmp_match_dist(const void *a, const void *b) {
  const orc_corr *x = (const orc_corr *)a, *y = (const orc_corr *)b;
  if (x->index_match != y->index_match) return x->index_match < y->index_match ? -1 : 1;
  if (x->distance != y->distance) return x->distance < y->distance ? -1 : 1;
  return (x->index_query > y->index_query) - (x->index_query < y->index_query);
}
static int cmp_dist(const void *a, const void *b) {
  const orc_corr *x = (const orc_corr *)a, *y = (const orc_corr *)b;
  if (x->distance != y->distance) return x->distance < y->distance ? -1 : 1;
  return (x->index_query > y->index_query) - (x->index_query < y->index_query);
}

int64_t orc_reject_one_to_one(const orc_corr *in, int64_t n, orc_corr *out) {
  orc_corr *t = (orc_corr *)malloc((size_t)(n ? n : 1) * sizeof(orc_corr));
  memcpy(t, in, (size_t)n * sizeof(orc_corr));
  qsort(t, (size_t)n, sizeof(orc_corr), cmp_match_dist); /* _one_to_one.hpp:50 */
  int64_t c = 0;
  int32_t last = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (t[i].index_match < 0) continue;
    if (t[i].index_match != last) {
      out[c++] = t[i];
      last = t[i].index_match;
    }
  }
  free(t);
  return c;
}

int64_t orc_reject_trimmed(const orc_corr *in, int64_t n, float overlap, unsigned min_corr,
                           orc_corr *out) {
  memcpy(out, in, (size_t)n * sizeof(orc_corr));
  unsigned keep = (unsigned)(int)floorf(overlap * (float)n); /* _trimmed.hpp:49 */
  if (keep < min_corr) keep = min_corr;
  if ((int64_t)keep < n) {
    qsort(out, (size_t)n, sizeof(orc_corr), cmp_dist);
    return keep;
  }
  return n;
}

/* correspondence_rejection_var_trimmed.hpp:44-107.  optimizeInlierRatio sorts the distances and
 * takes the argmin of FRMS_i = (1 / ratio_i^lambda)^2 * (1 / id_i) * (lower_sum + sorted[min_el + i])
 * (elementwise, as the reference's Eigen expression reads - not a cumulative sum); the kept set is
 * distance < sorted[int(n * factor)].
 * by_distance == 0: AS SHIPPED - the selection loop (:69-76) reads dists[i] of the scratch array that
 * optimizeInlierRatio sorted in place (:87), so original[i] is kept iff sorted[i] < trimmed: the first
 * #(sorted < trimmed) inputs.  by_distance != 0: each correspondence's own distance (the documented intent). */
int64_t orc_reject_var_trimmed(const orc_corr *in, int64_t n, double min_ratio, double max_ratio,
                               double lambda, int by_distance, orc_corr *out, double *factor_out,
                               double *trimmed_out) {
  if (factor_out) *factor_out = 0.0;
  if (trimmed_out) *trimmed_out = 0.0;
  if (n == 0) return 0;
  double *d = (double *)malloc((size_t)n * sizeof(double));
  for (int64_t i = 0; i < n; ++i) d[i] = in[i].distance;
  qsort(d, (size_t)n, sizeof(double), cmp_double);
  unsigned nbr = (unsigned)n;
  int min_el = (int)floor(min_ratio * nbr), max_el = (int)floor(max_ratio * nbr);
  double lower_sum = 0.0;
  for (int i = 0; i < min_el; ++i) lower_sum += d[i];
  int min_index = 0;
  double best = 0.0;
  for (int i = 0; i < max_el - min_el; ++i) {
    double id = (double)(min_el + 1 + i); /* LinSpaced(max_el - min_el, min_el + 1, max_el) */
    double ratio = id / (double)nbr;
    double deno = pow(ratio, lambda);
    double inv = 1.0 / deno;
    double frms = inv * inv * (1.0 / id) * (lower_sum + d[min_el + i]);
    if (i == 0 || frms < best) {
      best = frms;
      min_index = i;
    }
  }
  float opt_ratio = (float)(min_index + min_el) / (float)nbr;
  double factor = (double)opt_ratio;
  int64_t pos = (int64_t)(int)((double)n * factor);
  if (pos >= n) pos = n - 1;
  double trimmed = d[pos];
  if (factor_out) *factor_out = factor;
  if (trimmed_out) *trimmed_out = trimmed;
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if ((by_distance ? (double)in[i].distance : d[i]) < trimmed) out[c++] = in[i]; /* :72 */
  free(d);
  return c;
}

/* ------------------------------------------------------------------------------------ */
/* boost::mt19937 seeded 12345 behind uniform_int<>(0, INT_MAX) (sac_model.h:87-97):
 * with a 32-bit engine and a 2^31 range Boost divides the engine output into buckets of
 * two, i.e. rnd() == mt19937() >> 1.                                                     */
typedef struct {
  uint32_t mt[624];
  int idx;
} mt19937;
static void mt_seed(mt19937 *g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* eigenvalues of a symmetric 3x3 (stands in for pcl::eigen33; cyclic Jacobi in double) */
static void sym_eigvals3(const double Min[9], double ev[3]) {
  double a[9];
  memcpy(a, Min, sizeof(a));
  for (int sweep = 0; sweep < 50; ++sweep) {
    double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = a[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = a[k * 3 + p], akq = a[k * 3 + q];
          a[k * 3 + p] = c * akp - s * akq;
          a[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = a[p * 3 + k], aqk = a[q * 3 + k];
          a[p * 3 + k] = c * apk - s * aqk;
          a[q * 3 + k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = a[0];
  ev[1] = a[4];
  ev[2] = a[8];
}

/* sac_model_registration.h:257-279 + centroid.hpp:553-617 (fp32 accumulators) */
static double sample_dist_threshold(const float *src, int ss, const int32_t *sidx, int64_t n) {
  float acc[9] = {0};
  for (int64_t i = 0; i < n; ++i) {
    const float *p = src + (int64_t)sidx[i] * ss;
    acc[0] += p[0] * p[0];
    acc[1] += p[0] * p[1];
    acc[2] += p[0] * p[2];
    acc[3] += p[1] * p[1];
    acc[4] += p[1] * p[2];
    acc[5] += p[2] * p[2];
    acc[6] += p[0];
    acc[7] += p[1];
    acc[8] += p[2];
  }
  for (int a = 0; a < 9; ++a) acc[a] /= (float)n;
  float c00 = acc[0] - acc[6] * acc[6], c01 = acc[1] - acc[6] * acc[7],
        c02 = acc[2] - acc[6] * acc[8], c11 = acc[3] - acc[7] * acc[7],
        c12 = acc[4] - acc[7] * acc[8], c22 = acc[5] - acc[8] * acc[8];
  double M[9] = {c00, c01, c02, c01, c11, c12, c02, c12, c22}, ev[3];
  sym_eigvals3(M, ev);
  float s = 0.f;
  for (int a = 0; a < 3; ++a) s += sqrtf((float)(ev[a] > 0 ? ev[a] : 0));
  double thr = (double)s / 3.0;
  return thr * thr;
}

/* (T * (s,1) - (t,1)).squaredNorm() in fp32 (impl/sac_model_registration.hpp:211-220) */
static inline float model_sqdist(const float T[16], const float *s, const float *t) {
  float o[3];
  xform_point(T, s, o);
  float dx = o[0] - t[0], dy = o[1] - t[1], dz = o[2] - t[2];
  return (dx * dx + dy * dy) + dz * dz;
}

int64_t orc_ransac_registration(const float *src, int64_t ns, int ss, const float *tgt,
                                int64_t nt, int ts, const int32_t *sidx, const int32_t *tidx,
                                int64_t n, double inlier_threshold, int max_iterations,
                                int32_t *inlier_pos, float Tout[16], int *trials) {
  (void)ns;
  (void)nt;
  if (trials) *trials = 0;
  if (n < 3) return -1; /* getSamples clears the selection, computeModel breaks: ransac.hpp:72-76 */
  mt19937 g;
  mt_seed(&g, 12345u);
  double sample_thr = sample_dist_threshold(src, ss, sidx, n);
  /* shuffled_indices_ holds SOURCE indices; keep the position alongside so the target
   * index can be looked up without the reference's unordered_map (sidx is unique here) */
  int32_t *shuf = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  for (int64_t i = 0; i < n; ++i) shuf[i] = (int32_t)i; /* positions */
  double thresh = inlier_threshold * inlier_threshold;
  int iterations = 0, best = -INT_MAX, have_model = 0;
  double k = 1.0;
  unsigned skipped = 0;
  const unsigned max_skip = (unsigned)max_iterations * 10u;
  float Tbest[16];
  mat4_identity(Tbest);
  while (iterations < k && skipped < max_skip) {
    int sel[3], good = 0;
    for (unsigned iter = 0; iter < 1000u; ++iter) { /* max_sample_checks_, sac_model.h:470 */
      for (int i = 0; i < 3; ++i) {                  /* drawIndexSample :401-412 */
        uint32_t r = mt_next(&g) >> 1;
        int64_t j = i + (int64_t)(r % (uint32_t)(n - i));
        int32_t tmp = shuf[i];
        shuf[i] = shuf[j];
        shuf[j] = tmp;
      }
      for (int i = 0; i < 3; ++i) sel[i] = shuf[i];
      const float *p0 = src + (int64_t)sidx[sel[0]] * ss, *p1 = src + (int64_t)sidx[sel[1]] * ss,
                  *p2 = src + (int64_t)sidx[sel[2]] * ss;
      float a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      float b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      float c[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      if ((a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) > sample_thr && /* isSampleGood :47-60 */
          (b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) > sample_thr &&
          (c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) > sample_thr) {
        good = 1;
        break;
      }
    }
    if (!good) break; /* selection.empty(): ransac.hpp:72-76 */
    int32_t s3[3] = {sidx[sel[0]], sidx[sel[1]], sidx[sel[2]]};
    int32_t t3[3] = {tidx[sel[0]], tidx[sel[1]], tidx[sel[2]]};
    float T[16];
    orc_estimate_rigid_svd(src, ss, s3, tgt, ts, t3, 3, T);
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i)
      if ((double)model_sqdist(T, src + (int64_t)sidx[i] * ss, tgt + (int64_t)tidx[i] * ts) < thresh)
        ++cnt;
    if (cnt > best) {
      best = cnt;
      have_model = 1;
      memcpy(Tbest, T, sizeof(Tbest));
      double w = (double)best / (double)n;
      double p_no = 1.0 - pow(w, 3.0);
      if (p_no < DBL_EPSILON) p_no = DBL_EPSILON;
      if (p_no > 1.0 - DBL_EPSILON) p_no = 1.0 - DBL_EPSILON;
      k = log(1.0 - 0.99) / log(p_no);
    }
    ++iterations;
    if (iterations > max_iterations) break;
  }
  free(shuf);
  if (trials) *trials = iterations;
  if (!have_model) return -1;
  memcpy(Tout, Tbest, sizeof(Tbest));
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if ((double)model_sqdist(Tbest, src + (int64_t)sidx[i] * ss, tgt + (int64_t)tidx[i] * ts) < thresh)
      inlier_pos[c++] = (int32_t)i;
  return c;
}

/* ------------------------------------------------------------------------------------ */
void orc_icp_default_params(orc_icp_params *p) {
  p->max_iterations = 10;                          /* registration.h:97 */
  p->ransac_iterations = 1000;                     /* icp.h:104 */
  p->transformation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -DBL_MAX;
  p->max_correspondence_distance = sqrt(DBL_MAX);
  p->ransac_outlier_threshold = 0.05;
  p->min_correspondences = 3;
  p->svd_scalar_double = 0;
}

void orc_estimate_point_to_plane_lls(const float *src, int ss, const float *sn, int sns, const int32_t *sidx,
                                     const float *tgt, int ts, const float *tn, int tns, const int32_t *tidx,
                                     int64_t n, float T[16]);

/* tn != NULL: the loop runs with TransformationEstimationPointToPlaneLLS (registration.h:140-160 setTransformationEstimation,
 * icp.hpp:171); tn / sn are the normals of the target / source rows (sn nullable; the source normals are not moved by
 * transformPointCloud, icp.hpp:174, and the estimator only tests them for finiteness). */
static int icp_align_core(const orc_icp_params *p, const float *src, int64_t ns, int ss, const float *sn, int sns,
                          const float *tgt, int64_t nt, int ts, const float *tn, int tns, const float *guess,
                          float final_T[16], float *out, orc_icp_stats *stats, int nthreads, const float *scale) {
  float T[16], Tprev[16];
  mat4_identity(final_T);
  mat4_identity(T);
  mat4_identity(Tprev);
  if (stats) memset(stats, 0, sizeof(*stats));
  if (ns <= 0 || nt <= 0) return -1;
  /* registration.hpp:41-57: target copy with w = 1, tree on it */
  float *target = (float *)malloc((size_t)nt * 4 * sizeof(float));
  for (int64_t i = 0; i < nt; ++i) {
    memcpy(target + i * 4, tgt + i * ts, 12);
    target[i * 4 + 3] = 1.f;
  }
  /* setPointRepresentation (registration.h:362-370): the tree indexes vectorize(target point) = alpha * (x, y, z)
   * (point_representation.h:141-168, impl/kdtree_flann.hpp:216-244), the searches vectorize the moved source point
   * the same way (kdtree.h:204-218, impl/kdtree_flann.hpp:91-92); everything else reads the points themselves. */
  float *vtarget = target, *vq = NULL;
  if (scale) {
    vtarget = (float *)malloc((size_t)nt * 4 * sizeof(float));
    vq = (float *)malloc((size_t)ns * 4 * sizeof(float));
    for (int64_t i = 0; i < nt; ++i) {
      for (int c = 0; c < 3; ++c) vtarget[i * 4 + c] = target[i * 4 + c] * scale[c];
      vtarget[i * 4 + 3] = 1.f;
    }
  }
  orc_tree *tree = orc_tree_build(vtarget, nt, 4, 15);
  /* registration.hpp:149-163: output <- input, w = 1 */
  for (int64_t i = 0; i < ns; ++i) {
    memcpy(out + i * 4, src + i * ss, 12);
    out[i * 4 + 3] = 1.f;
  }
  int is_identity = 1;
  if (guess) {
    float I[16];
    mat4_identity(I);
    is_identity = memcmp(I, guess, sizeof(I)) == 0;
  }
  if (!is_identity) { /* icp.hpp:60-66 */
    memcpy(final_T, guess, 16 * sizeof(float));
    orc_transform(out, ns, 4, guess);
  }
  double dist_threshold = p->max_correspondence_distance * p->max_correspondence_distance;
  float *dcur = (float *)calloc((size_t)ns, sizeof(float));
  float *dprev = (float *)calloc((size_t)ns, sizeof(float));
  int32_t *nn_i = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  float *nn_d = (float *)malloc((size_t)ns * sizeof(float));
  int32_t *si = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  int32_t *ti = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  int32_t *inl = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  int32_t *sg = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  int32_t *tg = (int32_t *)malloc((size_t)ns * sizeof(int32_t));
  int converged = 0, iters = 0, rc = 0;
  int64_t cnt = 0;
  double fit = 0.0;
  while (!converged) { /* icp.hpp:72 */
    memcpy(Tprev, T, sizeof(T));
    memcpy(dprev, dcur, (size_t)ns * sizeof(float));
    if (scale) {
      for (int64_t i = 0; i < ns; ++i)
        for (int c = 0; c < 3; ++c) vq[i * 4 + c] = out[i * 4 + c] * scale[c];
    }
    orc_tree_knn(tree, scale ? vq : out, ns, 4, 1, nn_i, nn_d, nthreads);
    cnt = 0;
    for (int64_t i = 0; i < ns; ++i) {
      if ((double)nn_d[i] < dist_threshold) { /* strict: icp.hpp:93 */
        si[cnt] = (int32_t)i;
        ti[cnt] = nn_i[i];
        ++cnt;
      }
      float thr = (float)dist_threshold;
      dcur[i] = nn_d[i] < thr ? nn_d[i] : thr; /* icp.hpp:101 */
    }
    if (cnt < p->min_correspondences) { rc = -2; break; } /* icp.hpp:103-108 */
    int64_t ngood = cnt;
    const int32_t *sgood = si, *tgood = ti;
    if (p->ransac_iterations > 0) { /* icp.hpp:113-150 */
      float Tr[16];
      int64_t ni = orc_ransac_registration(out, ns, 4, target, nt, 4, si, ti, cnt,
                                           p->ransac_outlier_threshold, p->ransac_iterations,
                                           inl, Tr, NULL);
      if (ni >= 0) {
        for (int64_t a = 0; a < ni; ++a) {
          sg[a] = si[inl[a]];
          tg[a] = ti[inl[a]];
        }
        ngood = ni;
        sgood = sg;
        tgood = tg;
      }
    }
    if (ngood < p->min_correspondences) { rc = -2; break; } /* icp.hpp:153-159 */
    if (tn) /* icp.hpp:171 with the point-to-plane estimator */
      orc_estimate_point_to_plane_lls(out, 4, sn, sns, sgood, target, 4, tn, tns, tgood, ngood, T);
    else if (p->svd_scalar_double) /* icp.hpp:171 */
      orc_estimate_rigid_svd_f64(out, 4, sgood, target, 4, tgood, ngood, T);
    else
      orc_estimate_rigid_svd(out, 4, sgood, target, 4, tgood, ngood, T);
    orc_transform(out, ns, 4, T);                                     /* icp.hpp:174 */
    mat4_mul(T, final_T, final_T);                                    /* icp.hpp:177 */
    ++iters;
    float dsum = 0.f;
    for (int a = 0; a < 16; ++a) dsum += fabsf(T[a] - Tprev[a]);
    float fs = 0.f; /* registration.hpp:60-68: fp32 sum of (a - b), then / n in double */
    for (int64_t i = 0; i < ns; ++i) fs += dcur[i] - dprev[i];
    fit = (double)fs / (double)ns;
    if (iters >= p->max_iterations || (double)dsum < p->transformation_epsilon ||
        fabs(fit) <= p->euclidean_fitness_epsilon) /* icp.hpp:193-196 */
      converged = 1;
  }
  if (stats) {
    stats->iterations = iters;
    stats->converged = converged;
    stats->last_correspondences = (int)cnt;
    stats->fitness_delta = fit;
  }
  free(dcur); free(dprev); free(nn_i); free(nn_d); free(si); free(ti); free(inl); free(sg); free(tg);
  orc_tree_free(tree);
  if (scale) { free(vtarget); free(vq); }
  free(target);
  return rc;
}

int orc_icp_align_scaled(const orc_icp_params *p, const float scale[3], const float *src, int64_t ns, int ss,
                         const float *tgt, int64_t nt, int ts, const float *guess, float final_T[16], float *out,
                         orc_icp_stats *stats, int nthreads) {
  return icp_align_core(p, src, ns, ss, NULL, 0, tgt, nt, ts, NULL, 0, guess, final_T, out, stats, nthreads, scale);
}

int orc_icp_align(const orc_icp_params *p, const float *src, int64_t ns, int ss, const float *tgt,
                  int64_t nt, int ts, const float *guess, float final_T[16], float *out,
                  orc_icp_stats *stats, int nthreads) {
  return icp_align_core(p, src, ns, ss, NULL, 0, tgt, nt, ts, NULL, 0, guess, final_T, out, stats, nthreads, NULL);
}

int orc_icp_align_point_to_plane(const orc_icp_params *p, const float *src, int64_t ns, int ss, const float *sn,
                                 int sns, const float *tgt, int64_t nt, int ts, const float *tn, int tns,
                                 const float *guess, float final_T[16], float *out, orc_icp_stats *stats,
                                 int nthreads) {
  if (!tn) return -1;
  return icp_align_core(p, src, ns, ss, sn, sns, tgt, nt, ts, tn, tns, guess, final_T, out, stats, nthreads, NULL);
}

double orc_fitness_score(const float *src, int64_t ns, int ss, const float *tgt, int64_t nt,
                         int ts, const float final_T[16], double max_range) {
  orc_tree *tree = orc_tree_build(tgt, nt, ts, 15);
  double score = 0.0;
  int64_t nr = 0;
  for (int64_t i = 0; i < ns; ++i) {
    float q[3];
    xform_point(final_T, src + i * ss, q);
    int32_t j;
    float d;
    orc_tree_knn(tree, q, 1, 3, 1, &j, &d, 1);
    if ((double)d > max_range) continue; /* registration.hpp:94 */
    const float *t = tgt + (int64_t)j * ts;
    float dx = q[0] - t[0], dy = q[1] - t[1], dz = q[2] - t[2];
    score += fabs((double)((dx * dx + dy * dy) + dz * dz));
    ++nr;
  }
  orc_tree_free(tree);
  return nr > 0 ? score / (double)nr : DBL_MAX;
}

/* ===========================================================================================
 * Surface normals (features/include/pcl/features/normal_3d.h, impl/normal_3d.hpp)
 * =========================================================================================== */

/* computeRoots2 (common/include/pcl/common/eigen.h:105-116) */
static void eig_roots2(float b, float c, float roots[3]) {
  roots[0] = 0.f;
  float d = (float)(b * b - 4.0 * c);
  if (d < 0.0) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

/* computeRoots (eigen.h:122-185): closed-form eigenvalues of a symmetric 3x3, ascending, fp32 */
static void eig_roots3(const float m[9], float roots[3]) {
  float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
             m[8] * m[1] * m[1];
  float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920929e-07f) { /* one root is 0 -> quadratic */
    eig_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  float rho = sqrtf(-a_over_3);
  float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
  float cos_theta = cosf(theta), sin_theta = sinf(theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  }
  if (roots[0] <= 0.f) eig_roots2(c2, c1, roots);
}

/* eigen33 (mat, eigenvalue, eigenvector) (eigen.h:334-368): smallest eigenpair */
static void eigen33_smallest(const float mat[9], float *eigenvalue, float vec[3]) {
  float scale = 0.f;
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
  if (scale <= 1.17549435e-38f) scale = 1.f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float ev[3];
  eig_roots3(sm, ev);
  *eigenvalue = ev[0] * scale;
  sm[0] -= ev[0];
  sm[4] -= ev[0];
  sm[8] -= ev[0];
  const float *r0 = sm, *r1 = sm + 3, *r2 = sm + 6;
  float v1[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  float v2[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  float v3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float *best;
  float len;
  if (l1 >= l2 && l1 >= l3) { best = v1; len = l1; }
  else if (l2 >= l1 && l2 >= l3) { best = v2; len = l2; }
  else { best = v3; len = l3; }
  float s = sqrtf(len);
  for (int d = 0; d < 3; ++d) vec[d] = best[d] / s;
}

int orc_point_normal(const float *pts, int ps, const int32_t *idx, int64_t n, float plane[4],
                     float *curvature) {
  /* computeMeanAndCovarianceMatrix (centroid.hpp:553-617): nine fp32 running sums, then / n */
  float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float *p = pts + (size_t)(idx ? idx[i] : i) * ps;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    ++cnt;
    accu[0] += p[0] * p[0];
    accu[1] += p[0] * p[1];
    accu[2] += p[0] * p[2];
    accu[3] += p[1] * p[1];
    accu[4] += p[1] * p[2];
    accu[5] += p[2] * p[2];
    accu[6] += p[0];
    accu[7] += p[1];
    accu[8] += p[2];
  }
  if (cnt == 0) {
    plane[0] = plane[1] = plane[2] = plane[3] = *curvature = NAN;
    return 0;
  }
  for (int i = 0; i < 9; ++i) accu[i] /= (float)cnt;
  float cov[9];
  cov[0] = accu[0] - accu[6] * accu[6];
  cov[1] = accu[1] - accu[6] * accu[7];
  cov[2] = accu[2] - accu[6] * accu[8];
  cov[4] = accu[3] - accu[7] * accu[7];
  cov[5] = accu[4] - accu[7] * accu[8];
  cov[8] = accu[5] - accu[8] * accu[8];
  cov[3] = cov[1];
  cov[6] = cov[2];
  cov[7] = cov[5];
  /* solvePlaneParameters (feature.hpp:47-86) */
  float ev, vec[3];
  eigen33_smallest(cov, &ev, vec);
  plane[0] = vec[0];
  plane[1] = vec[1];
  plane[2] = vec[2];
  float eig_sum = cov[0] + cov[4] + cov[8];
  *curvature = eig_sum != 0.f ? fabsf(ev / eig_sum) : 0.f;
  plane[3] = -1.f * (plane[0] * accu[6] + plane[1] * accu[7] + plane[2] * accu[8]);
  return (int)cnt;
}

int orc_normals(const orc_tree *t, const float *pts, int ps, const float *q, int64_t nq, int qs, int k,
                double radius, const float vp[3], float *out, int nthreads) {
  const float r2 = (float)(radius * radius);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (int64_t i = 0; i < nq; ++i) {
    const float *qq = q + (size_t)i * qs;
    float *o = out + (size_t)i * 4;
    o[0] = o[1] = o[2] = o[3] = NAN;
    if (!isfinite(qq[0]) || !isfinite(qq[1]) || !isfinite(qq[2])) continue;
    int32_t *idx = NULL;
    float *d2 = NULL;
    int64_t m = 0;
    if (k > 0) {
      int ke = k > t->n ? (int)t->n : k;
      idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ke > 0 ? ke : 1));
      d2 = (float *)malloc(sizeof(float) * (size_t)(ke > 0 ? ke : 1));
      m = orc_tree_knn(t, qq, 1, qs, k, idx, d2, 1);
    } else {
      int64_t off[2];
      orc_tree_radius(t, qq, 1, qs, r2, 0, 1, off, &idx, &d2, 1);
      m = off[1];
    }
    if (m > 0) {
      float plane[4], curv;
      orc_point_normal(pts, ps, idx, m, plane, &curv);
      /* flipNormalTowardsViewpoint (normal_3d.h:162-181) */
      float vx = vp[0] - qq[0], vy = vp[1] - qq[1], vz = vp[2] - qq[2];
      float cos_theta = (vx * plane[0] + vy * plane[1] + vz * plane[2]);
      if (cos_theta < 0) {
        plane[0] *= -1;
        plane[1] *= -1;
        plane[2] *= -1;
      }
      o[0] = plane[0];
      o[1] = plane[1];
      o[2] = plane[2];
      o[3] = curv;
    }
    free(idx);
    free(d2);
  }
  return 0;
}

/* ===========================================================================================
 * Point-to-plane pieces
 * =========================================================================================== */

/* x = A^-1 b for a 6x6 by Gauss-Jordan with partial pivoting (stands in for Eigen's ATA.inverse () * ATb) */
static int solve6(double A[36], double b[6], double x[6]) {
  double M[6][7];
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c < 6; ++c) M[r][c] = A[r * 6 + c];
    M[r][6] = b[r];
  }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0.0) return -1;
    if (piv != c)
      for (int k = 0; k < 7; ++k) { double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      double f = M[r][c] / M[c][c];
      for (int k = c; k < 7; ++k) M[r][k] -= f * M[c][k];
    }
  }
  for (int r = 0; r < 6; ++r) x[r] = M[r][6] / M[r][r];
  return 0;
}

void orc_estimate_point_to_plane_lls(const float *src, int ss, const float *sn, int sns, const int32_t *sidx,
                                     const float *tgt, int ts, const float *tn, int tns, const int32_t *tidx,
                                     int64_t n, float T[16]) {
  double ATA[36], ATb[6];
  memset(ATA, 0, sizeof(ATA));
  memset(ATb, 0, sizeof(ATb));
  for (int64_t i = 0; i < n; ++i) {
    const size_t si = (size_t)(sidx ? sidx[i] : i), ti = (size_t)(tidx ? tidx[i] : i);
    const float *s = src + si * ss, *d = tgt + ti * ts, *nn = tn + ti * tns;
    int ok = isfinite(s[0]) && isfinite(s[1]) && isfinite(s[2]) && isfinite(d[0]) && isfinite(d[1]) &&
             isfinite(d[2]) && isfinite(nn[0]) && isfinite(nn[1]) && isfinite(nn[2]);
    if (ok && sn) {
      const float *m = sn + si * sns;
      ok = isfinite(m[0]) && isfinite(m[1]) && isfinite(m[2]);
    }
    if (!ok) continue;
    const float sx = s[0], sy = s[1], sz = s[2], dx = d[0], dy = d[1], dz = d[2];
    const float nx = nn[0], ny = nn[1], nz = nn[2];
    double a = nz * sy - ny * sz; /* fp32 expressions widened on assignment, as the reference writes them */
    double b = nx * sz - nz * sx;
    double c = ny * sx - nx * sy;
    ATA[0] += a * a;  ATA[1] += a * b;  ATA[2] += a * c;  ATA[3] += a * nx;  ATA[4] += a * ny;  ATA[5] += a * nz;
    ATA[7] += b * b;  ATA[8] += b * c;  ATA[9] += b * nx; ATA[10] += b * ny; ATA[11] += b * nz;
    ATA[14] += c * c; ATA[15] += c * nx; ATA[16] += c * ny; ATA[17] += c * nz;
    ATA[21] += nx * nx; ATA[22] += nx * ny; ATA[23] += nx * nz;
    ATA[28] += ny * ny; ATA[29] += ny * nz;
    ATA[35] += nz * nz;
    double dd = nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz;
    ATb[0] += a * dd; ATb[1] += b * dd; ATb[2] += c * dd; ATb[3] += nx * dd; ATb[4] += ny * dd; ATb[5] += nz * dd;
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < r; ++c) ATA[r * 6 + c] = ATA[c * 6 + r];
  double x[6] = {0, 0, 0, 0, 0, 0};
  mat4_identity(T);
  if (solve6(ATA, ATb, x) != 0) return;
  const double al = x[0], be = x[1], ga = x[2];
  T[0] = (float)(cos(ga) * cos(be));
  T[1] = -(float)(sin(ga) * cos(al) + cos(ga) * sin(be) * sin(al));
  T[2] = (float)(sin(ga) * sin(al) + cos(ga) * sin(be) * cos(al));
  T[4] = (float)(sin(ga) * cos(be));
  T[5] = (float)(cos(ga) * cos(al) + sin(ga) * sin(be) * sin(al));
  T[6] = -(float)(cos(ga) * sin(al) + sin(ga) * sin(be) * cos(al));
  T[8] = -(float)(sin(be));
  T[9] = (float)(cos(be) * sin(al));
  T[10] = (float)(cos(be) * cos(al));
  T[3] = (float)x[3];
  T[7] = (float)x[4];
  T[11] = (float)x[5];
  T[15] = 1.f;
}

int64_t orc_correspondences_normal_shooting(const float *src, int64_t ns, int ss, const float *sn, int sns,
                                            const float *tgt, int64_t nt, int ts, int k, double max_distance,
                                            orc_corr *out) {
  if (nt == 0 || k <= 0) return 0;
  orc_tree *t = orc_tree_build(tgt, nt, ts, 15);
  int ke = k > nt ? (int)nt : k;
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)ke);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)ke);
  int64_t cnt = 0;
  int min_index = 0; /* not reset per point in the reference either (:71) */
  for (int64_t i = 0; i < ns; ++i) {
    const float *p = src + (size_t)i * ss, *nrm = sn + (size_t)i * sns;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int m = orc_tree_knn(t, p, 1, ss, k, idx, d2, 1);
    float min_dist = FLT_MAX;
    for (int j = 0; j < m; ++j) {
      const float *q = tgt + (size_t)idx[j] * ts;
      const float vx = q[0] - p[0], vy = q[1] - p[1], vz = q[2] - p[2];
      const double N[3] = {nrm[0], nrm[1], nrm[2]}, V[3] = {vx, vy, vz};
      const double C[3] = {N[1] * V[2] - N[2] * V[1], N[2] * V[0] - N[0] * V[2], N[0] * V[1] - N[1] * V[0]};
      const float dist = (float)(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
      if (dist < min_dist) {
        min_dist = dist;
        min_index = j;
      }
    }
    if (min_dist > max_distance) continue;
    out[cnt].index_query = (int32_t)i;
    out[cnt].index_match = idx[min_index];
    out[cnt].distance = d2[min_index];
    ++cnt;
  }
  free(idx);
  free(d2);
  orc_tree_free(t);
  return cnt;
}

int64_t orc_correspondences_back_projection(const float *src, int64_t ns, int ss, const float *sn, int sns,
                                            const float *tgt, int64_t nt, int ts, const float *tn, int tns, int k,
                                            double max_distance, orc_corr *out) {
  if (nt == 0 || k <= 0) return 0;
  orc_tree *t = orc_tree_build(tgt, nt, ts, 15);
  int ke = k > nt ? (int)nt : k;
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)ke);
  float *d2 = (float *)malloc(sizeof(float) * (size_t)ke);
  int64_t cnt = 0;
  int min_index = 0; /* :71 - outside the loop over the points */
  for (int64_t i = 0; i < ns; ++i) {
    const float *p = src + (size_t)i * ss, *a = sn + (size_t)i * sns;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int m = orc_tree_knn(t, p, 1, ss, k, idx, d2, 1);
    float min_dist = FLT_MAX;
    for (int j = 0; j < m; ++j) {
      const float *b = tn + (size_t)idx[j] * tns;
      const float cos_angle = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];     /* :80-82 */
      const float dist = d2[min_index] * (2.0f - cos_angle * cos_angle);      /* :84 */
      if (dist < min_dist) {
        min_dist = dist;
        min_index = j;
      }
    }
    if (min_dist > max_distance) continue; /* :91-92 */
    out[cnt].index_query = (int32_t)i;
    out[cnt].index_match = idx[min_index];
    out[cnt].distance = d2[min_index];
    ++cnt;
  }
  free(idx);
  free(d2);
  orc_tree_free(t);
  return cnt;
}
