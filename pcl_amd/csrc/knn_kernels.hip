// knn_kernels.hip - exact k-nearest-neighbour search on gfx950 (MI355X), wave64.
//
// Replaces the FLANN traversal behind pcl::KdTreeFLANN::nearestKSearch
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:97-99).
//
// Design (DESIGN.md "kNN kernel"): queries are Morton-sorted, so the 64 queries of a
// wavefront are spatial neighbours.  A wavefront therefore walks the tree ONCE for all of its
// lanes ("union traversal"): the node stack is wave-uniform (one VGPR used as a 64-entry lane
// stack, read with v_readlane), node boxes and bucket points are fetched with SCALAR loads
// (s_load_dwordx16 through the constant address space) and live in SGPRs, and every lane
// tests every fetched point against its own query.  A node is visited iff ANY lane still needs
// it (its box distance <= that lane's current k-th distance).  There is no divergent memory
// access and no divergent control flow in the hot loop.
//
// k-best bookkeeping per lane:
//   * a[K]   sorted list of the K smallest DISTANCES seen so far, kept in registers and updated
//            with a branch-free v_med3_f32 chain: new a[j] = med3(a[j-1], a[j], d).
//            a[K-1] is the exact running k-th distance (the pruning bound).
//   * buf    per-lane append buffer in LDS holding the INDEX of every candidate that was <= the
//            bound when seen (4 bytes per entry: LDS per wave decides how many waves are resident,
//            and the traversal is latency-bound).  When a lane's buffer fills, the wave compacts it
//            against the current bound, recomputing the distances from the cloud rows.  At the end
//            the buffer is filtered with the final bound and the <= K survivors are sorted by the
//            64-bit key (d2 bits << 32 | index) in registers.
// This keeps the per-candidate cost at ~4 (packed distance) + K (med3) + ~4 (append) VALU
// instructions and still yields the exact (d2, index) order including ties.
#include <hipcub/hipcub.hpp>

// append-buffer capacities (entries per lane) of the index-only kernels, tuned on MI355X at 1M x 1M:
// small enough for 13-26 resident waves per CU, large enough to keep compactions rare
#ifndef PCLMI_IX8
#define PCLMI_IX8 24
#endif
#ifndef PCLMI_IX16
#define PCLMI_IX16 40
#endif
// (round 3: 48 entries = 12 KB per wave = three waves per SIMD instead of two with 80; 1M x 1M, k = 32: 1.047 -> 0.962 ms,
//  12.5M queries on the 10M-point target 12.42 -> 10.83 ms; 44 / 52 entries and four waves with 40: slower)
#ifndef PCLMI_IX32
#define PCLMI_IX32 48
#endif
// keep the pre-pass list (see knn_union_ix_kernel); 0 = round-1 behaviour, for A/B runs
#ifndef PCLMI_RETAIN
#define PCLMI_RETAIN 1
#endif
// epilogue with one gather per buffered candidate (knn_finish); 0 = filter, then gather the survivors again
// the per-lane (sparse-batch) kernel starts from an empty list instead of the pre-pass: its nearest-first descent
// reaches the query's own bucket anyway, and the pre-pass's second descent is pure latency there (k = 16: one query
// 33.7 -> 31.0 us, 16k queries 124 -> 118, 150k 211 -> 202; k = 32: 58.7 -> 49.7, 198 -> 190, 434 -> 421)
#ifndef PCLMI_LANE_NOPREPASS
#define PCLMI_LANE_NOPREPASS 1
#endif
#ifndef PCLMI_ONE_GATHER_MIN_K
#define PCLMI_ONE_GATHER_MIN_K 16
#endif
#ifndef PCLMI_NT_STORES
#define PCLMI_NT_STORES 0
#endif
#ifndef PCLMI_LEAF_RETEST
#define PCLMI_LEAF_RETEST 0
#endif
#ifndef PCLMI_FINISH_ONE_GATHER
#define PCLMI_FINISH_ONE_GATHER 1
#endif
#include "eig3.hpp"
#include "pclmi_internal.h"
#include "traverse.hpp"

namespace pclmi {
#ifdef PCLMI_STATS
// development counters (make variant NAME=stats EXTRA=-DPCLMI_STATS): per-launch totals over all waves
__device__ unsigned long long g_knn_stats[16];  // 0 buckets scanned, 1 candidates some lane appended, 2 chain executions,
                                                // 3 compactions, 4 inner expansions, 5 buckets popped, 6 waves,
                                                // 8 / 9 / 10 clock64 ticks of pre-pass / walk / epilogue (per wave)
extern "C" void pclmi_debug_knn_stats(unsigned long long out[16], int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_stats), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stats), z, sizeof(z));
  }
}
#define PCLMI_STAT(i, v)                                                  \
  do {                                                                    \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_knn_stats[i], (unsigned long long)(v)); \
  } while (0)
#define PCLMI_TICK(var) const long long var = clock64()
#else
#define PCLMI_TICK(var) \
  do {                  \
  } while (0)
#define PCLMI_STAT(i, v) \
  do {                   \
  } while (0)
#endif
namespace {

using namespace dev;

template <int N>
__device__ __forceinline__ void sort_keys(unsigned long long (&k)[N]) {
  // sorting network on N (power of two) 64-bit keys held in registers: Batcher's odd-even merge sort, a fifth fewer
  // exchanges than the bitonic sorter and all of them in one direction
  constexpr OddEvenMergeNet<N> net{};
#pragma unroll
  for (int c = 0; c < net.n; ++c) {
    const unsigned long long x = k[net.lo[c]], y = k[net.hi[c]];
    const bool sw = y < x;
    k[net.lo[c]] = sw ? y : x;
    k[net.hi[c]] = sw ? x : y;
  }
}

// ---------------------------------------------------------------------------------------------
// K >= 2.  The append buffer keeps 4 bytes per entry (the bucket slot of the point); distances are
// recomputed from the slot rows (x, y, z, idx) (same fp32 expression, so bit-identical) when the buffer is compacted and at the
// end.  Half the LDS per wave of a (d2, index) buffer -> twice the resident waves; the traversal is
// latency-bound, so occupancy is what it needs (DESIGN.md "kNN kernel").
// Compaction keeps entries with d <= tau (in place); if more than K survive (distance ties at the bound)
// the largest-index ties are dropped until exactly K remain - entries outside the current K best can
// never enter the final result.
template <int K, int C>
__device__ __forceinline__ int compact_index_buffer(unsigned int (*buf)[64], const float4 *__restrict__ cloud,
                                                    float qx, float qy, float qz, int lane, int cnt, float tau,
                                                    float (&dist_out)[K], unsigned int (&idx_out)[K], bool final) {
  int maxcnt = cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o));
  int n = 0;
  // eight entries per step: index reads, then eight independent row gathers in flight together
  for (int e = 0; e < maxcnt; e += 8) {
    unsigned int id[8];
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) id[u] = buf[min(e + u, C - 1)][lane];
#pragma unroll
    // (unconditional loads through a safe index: a load under a lane condition becomes a branch with its own
    // s_waitcnt, and the eight gathers run one after the other instead of together)
    for (int u = 0; u < 8; ++u) p[u] = cloud[(e + u < cnt) ? id[u] : 0u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float d = sq3(qx - p[u].x, qy - p[u].y, qz - p[u].z);
      if (e + u < cnt && d <= tau) {
        buf[n][lane] = id[u];
        ++n;
      }
    }
  }
  if (!final && n <= K) return n;
  // final pass (and the rare tie overflow): pull the <= C survivors' distances once more
  // ties at the bound: drop the largest-index ties until K remain
  while (n > K) {
    int worst = -1;
    unsigned int worst_idx = 0;
    for (int e = 0; e < n; ++e) {
      const float4 p = cloud[buf[e][lane]];
      const unsigned int id = __float_as_uint(p.w);  // external index
      const float d = sq3(qx - p.x, qy - p.y, qz - p.z);
      if (d == tau && (worst < 0 || id > worst_idx)) {
        worst = e;
        worst_idx = id;
      }
    }
    buf[worst][lane] = buf[n - 1][lane];
    --n;
  }
  if (final) {
    unsigned int sl[K];
#pragma unroll
    for (int j = 0; j < K; ++j) sl[j] = buf[min(j, C - 1)][lane];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float4 p = cloud[(j < n) ? sl[j] : 0u];
      dist_out[j] = sq3(qx - p.x, qy - p.y, qz - p.z);
      idx_out[j] = __float_as_uint(p.w);
    }
  }
  return n;
}

// Shared epilogue of the k-NN kernels: filter the lane's buffer with the exact k-th distance, sort the survivors by
// (d2, index), store the row.
template <int K, int C>
__device__ __forceinline__ void knn_finish(unsigned int (*buf)[64], const float4 *__restrict__ cloud, const LaneQuery &lq,
                                           int lane, int cnt, float tau, int k_out, int *__restrict__ out_idx,
                                           float *__restrict__ out_d2, int pitch) {
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  unsigned long long key[K];
  if constexpr (PCLMI_FINISH_ONE_GATHER && K >= PCLMI_ONE_GATHER_MIN_K && C / 2 >= K) {
  // ---- final: ONE gather per buffered candidate (K >= 16; below that the two-pass form is faster: 0.29 vs 0.26 ms
  // per 1M queries at K = 2, 0.41 vs 0.39 at K = 8, but 0.609 vs 0.622 at K = 16, 1.34 vs 1.36 at K = 32).  All bucket slots of the lane's buffer are read into registers
  // first, so the buffer's own LDS rows can take the survivors' (d2, index) keys (two rows per key) while the rows are
  // gathered eight at a time; the keys are then read back - no second gather of the K survivors.  More than K
  // survivors (distance ties at the bound; rare) put the slots back and take compact_index_buffer's tie handling.
  static_assert(C / 2 >= K, "the buffer must hold K keys");
  unsigned int id[C];
#pragma unroll
  for (int e = 0; e < C; ++e) id[e] = buf[e][lane];
  int maxcnt = cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o));
  int n = 0;
#pragma unroll
  for (int e0 = 0; e0 < C; e0 += 8) {
    if (e0 < maxcnt) {
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        p[u] = cloud[(e0 + u < C && e0 + u < cnt) ? id[e0 + u < C ? e0 + u : 0] : 0u];  // safe index: all eight in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float d = sq3(qx - p[u].x, qy - p[u].y, qz - p[u].z);
        if (e0 + u < C && e0 + u < cnt && d <= tau) {
          if (n < C / 2) {
            buf[2 * n][lane] = __float_as_uint(d);
            buf[2 * n + 1][lane] = __float_as_uint(p[u].w);  // external index
          }
          ++n;
        }
      }
    }
  }
  if (__any(n > K)) {
#pragma unroll
    for (int e = 0; e < C; ++e) buf[e][lane] = id[e];
    float dk[K];
    unsigned int ik[K];
    n = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, tau, dk, ik, true);
#pragma unroll
    for (int j = 0; j < K; ++j) key[j] = j < n ? (((unsigned long long)__float_as_uint(dk[j]) << 32) | ik[j]) : ~0ull;
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const unsigned int dj = buf[2 * j][lane], ij = buf[2 * j + 1][lane];
      key[j] = j < n ? (((unsigned long long)dj << 32) | ij) : ~0ull;
    }
  }
  } else {
  float dk[K];
  unsigned int ik[K];
  // ---- final: filter with the exact k-th distance, sort the survivors by (d2, index)
  cnt = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, tau, dk, ik, true);
#pragma unroll
  for (int j = 0; j < K; ++j) key[j] = j < cnt ? (((unsigned long long)__float_as_uint(dk[j]) << 32) | ik[j]) : ~0ull;
  }
  sort_keys<K>(key);
  if (lq.alive) {
    int *oi = out_idx + (size_t)lq.qi * pitch;
    float *od = out_d2 + (size_t)lq.qi * pitch;
    // full rows on 16-byte boundaries (the usual case): 128-bit stores - a 4-byte store per neighbour makes every
    // store instruction touch 64 different cache lines
    const bool wide = K % 4 == 0 && k_out == K && pitch % 4 == 0 &&
                      ((((unsigned long long)out_idx | (unsigned long long)out_d2) & 15ull) == 0ull);
    if (wide) {
#pragma unroll
      for (int j = 0; j + 3 < K; j += 4) {
        typedef int i4v __attribute__((ext_vector_type(4)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        const i4v vi = {(int)(unsigned int)key[j], (int)(unsigned int)key[j + 1], (int)(unsigned int)key[j + 2],
                        (int)(unsigned int)key[j + 3]};
        const f4v vd = {__uint_as_float((unsigned int)(key[j] >> 32)), __uint_as_float((unsigned int)(key[j + 1] >> 32)),
                        __uint_as_float((unsigned int)(key[j + 2] >> 32)), __uint_as_float((unsigned int)(key[j + 3] >> 32))};
#if PCLMI_NT_STORES
        // streaming stores: the 128 bytes per query are written once and not read here again - they should not push
        // the tree out of the XCD's L2
        __builtin_nontemporal_store(vi, reinterpret_cast<i4v *>(oi) + j / 4);
        __builtin_nontemporal_store(vd, reinterpret_cast<f4v *>(od) + j / 4);
#else
        reinterpret_cast<i4v *>(oi)[j / 4] = vi;
        reinterpret_cast<f4v *>(od)[j / 4] = vd;
#endif
      }
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j < k_out) {
          oi[j] = (int)(unsigned int)(key[j] & 0xffffffffull);
          od[j] = __uint_as_float((unsigned int)(key[j] >> 32));
        }
      }
    }
  }
}

// Epilogue of the FUSED surface-normal search (NormalEstimation::computeFeature, features/impl/normal_3d.hpp:46-98):
// instead of storing the k neighbours and gathering them again in a second kernel, the lane that has just found them
// accumulates the plane-fit moments (computeMeanAndCovarianceMatrix, common/impl/centroid.hpp:553-617; fp64, about
// the query point) from the bucket-slot rows its buffer points at - rows it touched a moment ago -, solves for the
// smallest eigenpair (solvePlaneParameters, features/impl/feature.hpp:47-86), flips the normal towards the viewpoint
// (normal_3d.h:162-181) and stores 16 bytes per query instead of 8 k.  The neighbour SET is the one pclmi_knn returns
// (the k smallest (d2, index) keys); the order of the fp64 additions differs from the two-kernel path's by
// roundings some 1e-9 below the fp32 result.
template <int K, int C>
__device__ __forceinline__ void normals_finish(unsigned int (*buf)[64], const float4 *__restrict__ cloud,
                                               const LaneQuery &lq, int lane, int cnt, float tau, int k_out, float vpx,
                                               float vpy, float vpz, float4 *__restrict__ out) {
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  unsigned long long kth = ~0ull;
  if (k_out < K) {
    // fewer neighbours asked for than the kernel keeps: the cut is the k_out-th smallest (d2, index) key
    float dk[K];
    unsigned int ik[K];
    cnt = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, tau, dk, ik, true);
    unsigned long long key[K];
#pragma unroll
    for (int j = 0; j < K; ++j) key[j] = j < cnt ? (((unsigned long long)__float_as_uint(dk[j]) << 32) | ik[j]) : ~0ull;
    sort_keys<K>(key);
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j == k_out - 1) kth = key[j];
  } else {
    float dk[K];
    unsigned int ik[K];
    cnt = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, tau, dk, ik, false);  // <= K survivors
  }
  double s1x = 0, s1y = 0, s1z = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0, n = 0;
#pragma unroll
  for (int j0 = 0; j0 < K; j0 += 8) {
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      p[u] = cloud[(j0 + u < K && j0 + u < cnt) ? buf[min(j0 + u, C - 1)][lane] : 0u];  // safe index: all eight in flight
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u >= K) continue;
      bool member = j0 + u < cnt;
      if (k_out < K) {
        const float d = sq3(qx - p[u].x, qy - p[u].y, qz - p[u].z);
        member = member && ((((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(p[u].w)) <= kth);
      }
      if (member) {
        const double dx = (double)p[u].x - (double)qx, dy = (double)p[u].y - (double)qy, dz = (double)p[u].z - (double)qz;
        n += 1.0;
        s1x += dx; s1y += dy; s1z += dz;
        xx += dx * dx; xy += dx * dy; xz += dx * dz;
        yy += dy * dy; yz += dy * dz; zz += dz * dz;
      }
    }
  }
  if (!lq.alive) return;
  const float nanf_ = __builtin_nanf("");
  float4 res = make_float4(nanf_, nanf_, nanf_, nanf_);
  if (lq.searchable && n > 0.0) {
    const double s1[3] = {s1x, s1y, s1z}, s2[6] = {xx, xy, xz, yy, yz, zz}, o[3] = {qx, qy, qz};
    double plane[4], curv;
    eig::plane_from_moments(n, s1, s2, o, plane, curv);
    const double c = ((double)vpx - qx) * plane[0] + ((double)vpy - qy) * plane[1] + ((double)vpz - qz) * plane[2];
    const double sgn = c < 0.0 ? -1.0 : 1.0;
    res = make_float4((float)(sgn * plane[0]), (float)(sgn * plane[1]), (float)(sgn * plane[2]), (float)curv);
  }
  out[lq.qi] = res;
}

// NRM: the fused surface-normal variant - out_d2 then receives nq x (nx, ny, nz, curvature), out_idx is unused
#ifndef PCLMI_WAVES16
#define PCLMI_WAVES16 4
#endif
#ifndef PCLMI_WAVES32
#define PCLMI_WAVES32 3
#endif
// LOW (k > 64, see launch_knn): one TILE of a long neighbour list - the K nearest among the points whose (d2, index)
// key lies ABOVE the last key of the previous tile, which the previous launch left at out[qi * pitch - 1] (out_idx /
// out_d2 point at the tile's first column; pitch = the row length of the whole list).
template <int K, int C, int WPB, bool NRM = false, bool LOW = false>
__global__ __launch_bounds__(64 * WPB)
    __attribute__((amdgpu_waves_per_eu(K == 16 ? PCLMI_WAVES16 : K <= 4 ? 7 : K == 8 ? 6 : K == 32 ? PCLMI_WAVES32 : 1))) void knn_union_ix_kernel(TreeView t, const float *__restrict__ q,
                                                                 int q_stride_f, const int *__restrict__ order,
                                                                 int nq, int k_out, int *__restrict__ out_idx,
                                                                 float *__restrict__ out_d2, float vpx, float vpy,
                                                                 float vpz, int qpw, int pitch) {
  static_assert(C >= K + 8, "buffer must hold the K best plus two groups of appends");
  static_assert(!(LOW && NRM), "tiles are neighbour lists");
  __shared__ unsigned int lds[WPB][C][64];
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<WPB>(q, q_stride_f, order, nq, lq, qpw)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  unsigned int(*buf)[64] = lds[threadIdx.x >> 6];
  const float4 *cloud = reinterpret_cast<const float4 *>(t.slot);  // bucket-slot rows (x, y, z, idx bits)

  PCLMI_TICK(tick0);
  float a[K];
#if PCLMI_RETAIN
  // The pre-pass list is KEPT: a[] starts as the K smallest distances of the lane's home buckets (real points the
  // walk will meet again), not K copies of their K-th distance, so the bound tightens with the first outside
  // candidate instead of after K of them.  When the walk reaches a home bucket the lane appends what is still within
  // its bound but inserts nothing (those distances are in a[] already).
  int home = -0x40000000;  // no bucket number is near it: lanes without a query have no home
  float lo_d = -1.f;        // LOW: the previous tile's last key; only keys above it count
  unsigned int lo_i = 0u;
  if (lq.searchable) {
    if constexpr (LOW) {
      // no pre-pass: its list would have to leave out the points of the earlier tiles; the walk starts unbounded
      lo_d = out_d2[(size_t)lq.qi * pitch - 1];
      lo_i = (unsigned int)out_idx[(size_t)lq.qi * pitch - 1];
#pragma unroll
      for (int j = 0; j < K; ++j) a[j] = 3.402823466e+38f;
    } else {
#if PCLMI_NOPREPASS
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = 3.402823466e+38f;
#else
    lane_bound_list<K>(t, qx, qy, qz, a, home);
#endif
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = -1.f;
  }
#else
  const float bound = lq.searchable ? lane_bound<K>(t, qx, qy, qz) : -1.f;
  asm volatile("" ::"v"(bound));
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = bound;
#endif
  int cnt = 0;
  float dk[K];
  unsigned int ik[K];

  LaneStack st{0, 0, lane};
  st.push(1);
#if PCLMI_LEAF_RETEST
  const cbox_ptr box = (cbox_ptr)t.box;
#endif
  const cgrp_ptr box2 = (cgrp_ptr)t.box;
  const cgrp_ptr leaf = (cgrp_ptr)t.leaf;
  const int M = t.M;
  asm volatile("" ::"v"(a[K - 1]));
  PCLMI_TICK(tick1);

  // Two phases per round, so that the k-best list is only ever WRITTEN in straight-line code that every round runs
  // (as one loop with a `continue` for pruned buckets the compiler keeps two copies of a[] and moves all K values
  // across at every bucket): pop and expand until a bucket some lane still needs comes up, then scan that bucket.
  for (;;) {
    int node = 0;
    bool found = false;
    while (st.sp > 0) {
      node = st.pop();
      if (node >= M) {
        PCLMI_STAT(5, 1);
#if PCLMI_LEAF_RETEST
        const f8v b = box[node];
        const float md = box_dist(qx, qy, qz, b[0], b[1], b[2], b[3], b[4], b[5]);
        if (__any(md <= a[K - 1])) {
          found = true;
          break;
        }
#else
        // no second box test: expand_inner tested this bucket's box when it pushed it, and the bounds have rarely
        // moved enough since to prune it (measured: 1.2 of 46 popped buckets per wavefront) - less than the 14
        // instructions and the scalar load the test costs on every bucket
        found = true;
        break;
#endif
      } else {
        PCLMI_STAT(4, 1);
        expand_inner(st, box2, node, M, qx, qy, qz, a[K - 1]);
      }
    }
    if (!found) break;
    {
      const cgrp_ptr rec = leaf + (size_t)(node - M) * G;
      f16v grp[G];
#pragma unroll
      for (int g = 0; g < G; ++g) grp[g] = rec[g];
      PCLMI_STAT(0, 1);
      const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#if PCLMI_RETAIN
      const bool is_home = (unsigned int)(node - home) < (unsigned int)prepass_buckets<K>();
      const unsigned long long homeb = __ballot(is_home);
#endif
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float dd[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f2v d2 = sq3_pk(q2x - f2v{grp[g][2 * h], grp[g][2 * h + 1]},
                                q2y - f2v{grp[g][4 + 2 * h], grp[g][5 + 2 * h]},
                                q2z - f2v{grp[g][8 + 2 * h], grp[g][9 + 2 * h]});
          dd[2 * h] = d2.x;
          dd[2 * h + 1] = d2.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = dd[j];
          bool pass = d <= a[K - 1];
          bool above = true;
          if constexpr (LOW) {
            const unsigned int ib = __float_as_uint(grp[g][12 + j]);
            above = d > lo_d || (d == lo_d && ib > lo_i);
            pass = pass && above;
          }
#if PCLMI_RETAIN
          const unsigned long long pb = __ballot(pass);
          if (pb) {
            PCLMI_STAT(1, 1);
            if (pass) {
              buf[cnt][lane] = (unsigned int)(node - M) * L + g * 4 + j;  // bucket slot
              ++cnt;
            }
            if (pb & ~homeb) {  // some lane meets this point for the first time
              PCLMI_STAT(2, 1);
              list_insert<K>(a, (is_home || !above) ? __builtin_inff() : d);
            }
          }
#else
          if (__any(pass)) {
            PCLMI_STAT(2, 1);
            if (pass) {
              buf[cnt][lane] = (unsigned int)(node - M) * L + g * 4 + j;  // bucket slot
              ++cnt;
            }
            list_insert<K>(a, d);
          }
#endif
        }
        if (__any(cnt > C - 4)) {
          PCLMI_STAT(3, 1);
          cnt = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, a[K - 1], dk, ik, false);
        }
      }
    }
  }
  PCLMI_STAT(6, 1);
  asm volatile("" ::"v"(a[K - 1]));
  PCLMI_TICK(tick2);

  if constexpr (NRM)
    normals_finish<K, C>(buf, cloud, lq, lane, cnt, a[K - 1], k_out, vpx, vpy, vpz, reinterpret_cast<float4 *>(out_d2));
  else
    knn_finish<K, C>(buf, cloud, lq, lane, cnt, a[K - 1], k_out, out_idx, out_d2, pitch);
#ifdef PCLMI_STATS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PCLMI_TICK(tick3);
  PCLMI_STAT(8, tick1 - tick0);
  PCLMI_STAT(9, tick2 - tick1);
  PCLMI_STAT(10, tick3 - tick2);
#endif
}

// ---------------------------------------------------------------------------------------------
// K >= 2, SPARSE batches (far fewer queries than target points): the 64 queries of a wavefront are then far apart,
// the union of their search balls is most of a subtree (10 000 queries on a 1M-point target: ~400 buckets per wave
// against ~10 per query), and every lane walking for itself (traverse.hpp lane_traverse) wins although its loads
// diverge.  Same bookkeeping as the union kernel: K running distances in registers, index-only append buffer in
// LDS, compaction when a bucket's worth of room is no longer free, shared epilogue.
template <int K, int C, bool NRM = false>
__global__ __launch_bounds__(64) void knn_lane_ix_kernel(TreeView t, const float *__restrict__ q, int q_stride_f,
                                                         const int *__restrict__ order, int nq, int levels, int k_out,
                                                         int *__restrict__ out_idx, float *__restrict__ out_d2,
                                                         float vpx, float vpy, float vpz, int qpw) {
  static_assert(C >= K + L + 8, "buffer must hold the K best plus one bucket of appends");
  extern __shared__ float far_lds[];  // levels x 64 floats
  __shared__ unsigned int lds[C][64];
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<1>(q, q_stride_f, order, nq, lq, qpw)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  unsigned int(*buf)[64] = lds;
  const float4 *cloud = reinterpret_cast<const float4 *>(t.slot);
  // (with PCLMI_LANE_NOPREPASS=0) the pre-pass list is kept, as in the union kernel: a[] starts as the K smallest
  // distances of the home buckets, and in a home bucket the lane appends what is within its bound but inserts nothing
  float a[K];
  int home = -0x40000000;
  if (lq.searchable) {
#if PCLMI_LANE_NOPREPASS
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = 3.402823466e+38f;
#else
    lane_bound_list<K>(t, qx, qy, qz, a, home);
#endif
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = -1.f;
  }
  const float bound = a[K - 1];
  int cnt = 0;
  float dk[K];
  unsigned int ik[K];
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  const int M_ = t.M;
  lane_traverse(
      t, qx, qy, qz, lq.searchable, bound, reinterpret_cast<float(*)[64]>(far_lds), lane, 1,
      [&](int leafno, float) {
        const bool is_home = (unsigned int)(leafno + M_ - home) < (unsigned int)prepass_buckets<K>();
        const float4 *rec = reinterpret_cast<const float4 *>(t.leaf + (size_t)leafno * (4 * L));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 X = rec[g * 4 + 0], Y = rec[g * 4 + 1], Z = rec[g * 4 + 2];
          const f2v d01 = sq3_pk(q2x - f2v{X.x, X.y}, q2y - f2v{Y.x, Y.y}, q2z - f2v{Z.x, Z.y});
          const f2v d23 = sq3_pk(q2x - f2v{X.z, X.w}, q2y - f2v{Y.z, Y.w}, q2z - f2v{Z.z, Z.w});
          const float dd[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (dd[j] <= a[K - 1]) {
              buf[cnt][lane] = (unsigned int)leafno * L + g * 4 + j;  // bucket slot
              ++cnt;
              list_insert<K>(a, is_home ? __builtin_inff() : dd[j]);
            }
          }
        }
        return a[K - 1];
      },
      [&] {  // once per round, wave converged: make room for the next bucket
        if (__any(cnt > C - L)) cnt = compact_index_buffer<K, C>(buf, cloud, qx, qy, qz, lane, cnt, a[K - 1], dk, ik, false);
      });
  if constexpr (NRM)
    normals_finish<K, C>(buf, cloud, lq, lane, cnt, a[K - 1], k_out, vpx, vpy, vpz, reinterpret_cast<float4 *>(out_d2));
  else
    knn_finish<K, C>(buf, cloud, lq, lane, cnt, a[K - 1], k_out, out_idx, out_d2, k_out);
}

// ---------------------------------------------------------------------------------------------
// K == 1 : running best 64-bit key, no buffer.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void knn1_union_kernel(TreeView t, const float *__restrict__ q,
                                                               int q_stride_f,
                                                               const int *__restrict__ order, int nq,
                                                               int *__restrict__ out_idx,
                                                               float *__restrict__ out_d2, int qpw) {
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<WPB>(q, q_stride_f, order, nq, lq, qpw)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  const unsigned long long best = wave_nearest(t, qx, qy, qz, lq.searchable, lane);
  if (lq.alive) {
    out_idx[lq.qi] = lq.searchable ? (int)(unsigned int)(best & 0xffffffffull) : -1;
    out_d2[lq.qi] = lq.searchable ? __uint_as_float((unsigned int)(best >> 32)) : __builtin_nanf("");
  }
}

// K == 1, per-lane traversal (traverse.hpp lane_nearest): dynamic LDS = WPB x levels x 64 floats
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void knn1_lane_kernel(TreeView t, const float *__restrict__ q,
                                                              int q_stride_f, const int *__restrict__ order,
                                                              int nq, int levels, int *__restrict__ out_idx,
                                                              float *__restrict__ out_d2, int qpw) {
  extern __shared__ float far_lds[];
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<WPB>(q, q_stride_f, order, nq, lq, qpw)) return;
  int best_slot;
  const unsigned long long best =
      lane_nearest(t, lq.x, lq.y, lq.z, lq.searchable, __builtin_inff(),
                   reinterpret_cast<float(*)[64]>(far_lds + (size_t)(threadIdx.x >> 6) * levels * 64), lane, 1,
                   best_slot);
  if (lq.alive) {
    out_idx[lq.qi] = lq.searchable ? (int)(unsigned int)(best & 0xffffffffull) : -1;
    out_d2[lq.qi] = lq.searchable ? __uint_as_float((unsigned int)(best >> 32)) : __builtin_nanf("");
  }
}

// ---------------------------------------------------------------------------------------------
// Any k: one query per lane, private traversal, (d2,index) max-heap in global scratch laid out
// [k][nq] so that neighbouring lanes touch neighbouring words.  Slow path for k > 32.
__global__ __launch_bounds__(64) void knn_heap_kernel(TreeView t, const float *__restrict__ q,
                                                      int q_stride_f, const int *__restrict__ order,
                                                      int nq, int k, unsigned long long *heap,
                                                      int *__restrict__ out_idx,
                                                      float *__restrict__ out_d2) {
  const int slot = blockIdx.x * 64 + threadIdx.x;
  if (slot >= nq) return;
  const int qi = order ? order[slot] : slot;
  const float *p = q + (size_t)qi * q_stride_f;
  const float qx = p[0], qy = p[1], qz = p[2];
  if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) {  // as every other k-NN kernel answers a query that is no point
    for (int j = 0; j < k; ++j) {
      out_idx[(size_t)qi * k + j] = -1;
      out_d2[(size_t)qi * k + j] = __builtin_nanf("");
    }
    return;
  }
  unsigned long long *h = heap + slot;  // element e at h[e * nq]
  const size_t S = (size_t)nq;
  int cnt = 0;
  int stack[64];
  int sp = 0;
  stack[sp++] = 1;
  const int M = t.M;
  while (sp > 0) {
    const int node = stack[--sp];
    const float *b = t.box + (size_t)node * 8;
    float dx = fmaxf(fmaxf(b[0] - qx, qx - b[3]), 0.f);
    float dy = fmaxf(fmaxf(b[1] - qy, qy - b[4]), 0.f);
    float dz = fmaxf(fmaxf(b[2] - qz, qz - b[5]), 0.f);
    const float md = sq3(dx, dy, dz);
    if (cnt == k && md > __uint_as_float((unsigned int)(h[0] >> 32))) continue;
    if (node >= M) {
      const float *rec = t.leaf + (size_t)(node - M) * (4 * L);
      for (int j = 0; j < L; ++j) {
        const float *gp = rec + (j >> 2) * 16 + (j & 3);
        const int idx = __float_as_int(gp[12]);
        if (idx < 0) continue;  // padded slot
        const float d = sq3(qx - gp[0], qy - gp[4], qz - gp[8]);
        const unsigned long long key = make_key(d, idx);
        if (cnt < k) {  // sift up
          int c = cnt++;
          while (c > 0) {
            int par = (c - 1) >> 1;
            unsigned long long pk = h[par * S];
            if (pk >= key) break;
            h[c * S] = pk;
            c = par;
          }
          h[c * S] = key;
        } else if (key < h[0]) {  // replace the maximum, sift down
          int c = 0;
          for (;;) {
            int l = 2 * c + 1, r = l + 1;
            if (l >= k) break;
            unsigned long long lk = h[l * S];
            int m = l;
            unsigned long long mk = lk;
            if (r < k) {
              unsigned long long rk = h[r * S];
              if (rk > lk) { m = r; mk = rk; }
            }
            if (mk <= key) break;
            h[c * S] = mk;
            c = m;
          }
          h[c * S] = key;
        }
      }
    } else {
      // visit the nearer child first
      const float *b0 = t.box + (size_t)node * 16;
      float ax = fmaxf(fmaxf(b0[0] - qx, qx - b0[3]), 0.f), ay = fmaxf(fmaxf(b0[1] - qy, qy - b0[4]), 0.f),
            az = fmaxf(fmaxf(b0[2] - qz, qz - b0[5]), 0.f);
      float bx = fmaxf(fmaxf(b0[8] - qx, qx - b0[11]), 0.f), by = fmaxf(fmaxf(b0[9] - qy, qy - b0[12]), 0.f),
            bz = fmaxf(fmaxf(b0[10] - qz, qz - b0[13]), 0.f);
      if (sq3(ax, ay, az) <= sq3(bx, by, bz)) {
        stack[sp++] = 2 * node + 1;
        stack[sp++] = 2 * node;
      } else {
        stack[sp++] = 2 * node;
        stack[sp++] = 2 * node + 1;
      }
    }
  }
  // heap sort: repeatedly move the maximum to the end -> ascending order
  for (int end = cnt - 1; end > 0; --end) {
    unsigned long long top = h[0], key = h[end * S];
    h[end * S] = top;
    int c = 0;
    for (;;) {
      int l = 2 * c + 1, r = l + 1;
      if (l >= end) break;
      unsigned long long lk = h[l * S];
      int m = l;
      unsigned long long mk = lk;
      if (r < end) {
        unsigned long long rk = h[r * S];
        if (rk > lk) { m = r; mk = rk; }
      }
      if (mk <= key) break;
      h[c * S] = mk;
      c = m;
    }
    h[c * S] = key;
  }
  int *oi = out_idx + (size_t)qi * k;
  float *od = out_d2 + (size_t)qi * k;
  for (int j = 0; j < k; ++j) {
    unsigned long long key = j < cnt ? h[j * S] : ~0ull;
    oi[j] = j < cnt ? (int)(unsigned int)(key & 0xffffffffull) : -1;
    od[j] = j < cnt ? __uint_as_float((unsigned int)(key >> 32)) : __builtin_nanf("");
  }
}

// ---------------------------------------------------------------------------------------------
// Morton codes of the queries relative to the tree's bounding box (10 bits per axis).
__device__ __forceinline__ unsigned int spread10(unsigned int v) {
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// 30-bit curve index of a 10-bit-per-axis grid cell; its top bits are the index of the coarser cell.
// PCLMI_HILBERT: Hilbert curve (Skilling's axes-to-transpose) instead of the Z-order: consecutive cells are always
// face neighbours, so the 64 consecutive queries of a wavefront never straddle one of the Z-order's long jumps.
#ifndef PCLMI_HILBERT
#define PCLMI_HILBERT 1
#endif
__device__ __forceinline__ unsigned int curve_index(unsigned int ix, unsigned int iy, unsigned int iz) {
#if PCLMI_HILBERT
  unsigned int X[3] = {ix, iy, iz};
#pragma unroll
  for (unsigned int Q = 512u; Q > 1u; Q >>= 1) {
    const unsigned int P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const unsigned int t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned int t = 0u;
#pragma unroll
  for (unsigned int Q = 512u; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  X[0] ^= t;
  X[1] ^= t;
  X[2] ^= t;
  return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
#else
  return spread10(ix) | (spread10(iy) << 1) | (spread10(iz) << 2);
#endif
}

__device__ __forceinline__ unsigned int morton_cell(const float *p, float lox, float loy, float loz, float sx, float sy,
                                                    float sz, int shift) {
  float fx = (p[0] - lox) * sx, fy = (p[1] - loy) * sy, fz = (p[2] - loz) * sz;
  // NaN / out-of-box queries clamp into the grid; order only affects speed, never results
  unsigned int ix = (unsigned int)fminf(fmaxf(fx, 0.f), 1023.f);
  unsigned int iy = (unsigned int)fminf(fmaxf(fy, 0.f), 1023.f);
  unsigned int iz = (unsigned int)fminf(fmaxf(fz, 0.f), 1023.f);
  return curve_index(ix, iy, iz) >> shift;
}

// "Tube" order for clouds that fill their bounding box (pclmi_tree::fill): the y-z plane is cut into gy x gz square
// tubes along x whose cross-section is the side of a cube that holds one wavefront's 64 queries, the tubes are walked
// boustrophedon (neighbouring tubes are face neighbours, odd tubes run backwards) and the queries of a tube are ordered
// by x.  ANY 64 consecutive queries are then a slab of a tube, about a cube - whereas 64 consecutive queries along the
// Hilbert curve are a chain of ~17 small cells of arbitrary shape.  Measured on the 1M x 1M uniform case, k = 16: the
// union of a wavefront's search balls touches 37 buckets instead of 47.
// key = tube << 10 | x bin (1024 bins, reversed in odd tubes)
// tube_p > 0: the tubes are numbered along a 2-D Hilbert curve over a tube_p x tube_p grid (a power of two >= gy, gz)
// instead of row by row: the ~20 tubes the wavefronts of one XCD work on at a time are then a compact patch of the
// y-z plane, not a strip across it, and the part of the tree they touch fits the XCD's L2 (FETCH_SIZE 139 -> 116 MB
// per launch, k = 16 kernel 0.538 -> 0.526 ms).
__device__ __forceinline__ unsigned int hilbert2d(unsigned int x, unsigned int y, unsigned int n) {
  unsigned int d = 0u;
  for (unsigned int s = n >> 1; s > 0u; s >>= 1) {
    const unsigned int rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (!ry) {
      if (rx) {
        x = n - 1u - x;
        y = n - 1u - y;
      }
      const unsigned int t = x;
      x = y;
      y = t;
    }
  }
  return d;
}

// tube_of: the tube numbers of the tube_p x tube_p grid (the count kernel tabulates hilbert2d once per workgroup, in LDS)
__device__ __forceinline__ unsigned int tube_key(const float *p, float lox, float loy, float loz, float sx, float sy,
                                                 float sz, int gy, int gz, int tube_p,
                                                 const unsigned int *tube_of) {
  const float fx = (p[0] - lox) * sx, fy = (p[1] - loy) * sy, fz = (p[2] - loz) * sz;
  const unsigned int xb = (unsigned int)fminf(fmaxf(fx, 0.f), 1023.f);  // NaN clamps to 0: order only affects speed
  const unsigned int ty = (unsigned int)fminf(fmaxf(fy, 0.f), (float)(gy - 1));
  const unsigned int tz = (unsigned int)fminf(fmaxf(fz, 0.f), (float)(gz - 1));
  const unsigned int tube = tube_p > 0 ? tube_of[tz * (unsigned int)tube_p + ty]
                                       : tz * (unsigned int)gy + ((tz & 1u) ? (unsigned int)gy - 1u - ty : ty);
  return (tube << 10) | ((tube & 1u) ? 1023u - xb : xb);
}

// (Measured and dropped: leaving the queries themselves in sorted order, as 16-byte records carried through the
// scatter or gathered by bin_sort_kernel, so that the search kernels load them coalesced - the k = 16 kernel gains 19 us
// (0.526 -> 0.507 ms; its FETCH falls by the 64-byte sector per 12-byte row), the ordering loses 21-25 us.)
// Query ordering = counting sort by Morton cell.  Only coherence matters (the order inside a
// cell is irrelevant and results never depend on it), so one histogram pass, one scan and one
// scatter pass replace a full radix sort.
__global__ void cell_count_kernel(const float *__restrict__ q, int q_stride_f, int nq, float lox, float loy,
                                  float loz, float sx, float sy, float sz, int shift,
                                  unsigned int *__restrict__ cell, unsigned int *__restrict__ rank,
                                  unsigned int *__restrict__ hist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const unsigned int c = morton_cell(q + (size_t)i * q_stride_f, lox, loy, loz, sx, sy, sz, shift);
  cell[i] = c;
  rank[i] = atomicAdd(&hist[c], 1u);  // position inside the cell: the scatter needs no second atomic
}

// curve index of every query (the stable ordering sorts them with a radix sort)
__global__ void cell_only_kernel(const float *__restrict__ q, int q_stride_f, int nq, float lox, float loy, float loz,
                                 float sx, float sy, float sz, int shift, unsigned int *__restrict__ cell) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) cell[i] = morton_cell(q + (size_t)i * q_stride_f, lox, loy, loz, sx, sy, sz, shift);
}

__global__ void cell_scatter_kernel(const unsigned int *__restrict__ cell, const unsigned int *__restrict__ rank,
                                    int nq, const unsigned int *__restrict__ offs, int *__restrict__ order) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  order[offs[cell[i]] + rank[i]] = i;
}


// ---- two-level counting sort (the default for large batches) ------------------------------------------------------
// cell_count_kernel above pays one device-wide returning atomic per query (43 us per 1M queries: they are resolved
// beyond the XCD's L2).  Here the cell number is split into a high and a low part and every counter lives in LDS:
//   1. tile_count_kernel: one workgroup per tile of the batch; LDS histogram of the HIGH part, the returning LDS
//      atomic is the query's rank inside (tile, high bin); the tile's histogram goes to tilehist[bin][tile];
//   2. exclusive scan of tilehist (bin-major): start of every (bin, tile) run, hence of every high bin;
//   3. tile_scatter_kernel: (low part, row) pairs to their high bin (order inside a bin: by tile, then rank);
//   4. bin_sort_kernel: one workgroup per high bin counting-sorts its run by the LOW part in LDS (histogram, scan,
//      returning LDS atomics for the position) and writes the row numbers to `order`.
// Any bin size works (a workgroup loops over its run); the order inside a cell is arbitrary, as before.
constexpr int TILE_TPB = 1024;
__global__ __launch_bounds__(TILE_TPB) void tile_count_kernel(const float *__restrict__ q, int q_stride_f, int nq, float lox,
                                                              float loy, float loz, float sx, float sy, float sz, int shift,
                                                              int lo_bits, int nb_hi, int tile_size, int ntiles, int gy, int gz,
                                                              int tube_p, unsigned int *__restrict__ cell,
                                                              unsigned int *__restrict__ rank,
                                                              unsigned int *__restrict__ tilehist) {
  extern __shared__ unsigned int lds_hist[];  // nb_hi counters, then the tube_p x tube_p tube numbers
  unsigned int *tube_of = lds_hist + nb_hi;
  for (int b = threadIdx.x; b < nb_hi; b += TILE_TPB) lds_hist[b] = 0u;
  for (int b = threadIdx.x; b < tube_p * tube_p; b += TILE_TPB)
    tube_of[b] = hilbert2d((unsigned int)(b % tube_p), (unsigned int)(b / tube_p), (unsigned int)tube_p);
  __syncthreads();
  const int tile = blockIdx.x;
  const long long begin = (long long)tile * tile_size;
  const long long end = begin + tile_size < (long long)nq ? begin + tile_size : (long long)nq;
  for (long long i = begin + threadIdx.x; i < end; i += TILE_TPB) {
    const unsigned int c = gy > 0 ? tube_key(q + (size_t)i * q_stride_f, lox, loy, loz, sx, sy, sz, gy, gz, tube_p, tube_of)
                                  : morton_cell(q + (size_t)i * q_stride_f, lox, loy, loz, sx, sy, sz, shift);
    cell[i] = c;
    rank[i] = atomicAdd(&lds_hist[c >> lo_bits], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb_hi; b += TILE_TPB) tilehist[(size_t)b * ntiles + tile] = lds_hist[b];
  if (tile == 0 && threadIdx.x == 0) tilehist[(size_t)nb_hi * ntiles] = 0u;  // the scan turns it into nq
}

__global__ void tile_scatter_kernel(const unsigned int *__restrict__ cell, const unsigned int *__restrict__ rank, int nq,
                                    int lo_bits, int tile_size, int ntiles, const unsigned int *__restrict__ scanned,
                                    uint2 *__restrict__ pairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const unsigned int c = cell[i];
  const unsigned int pos = scanned[(size_t)(c >> lo_bits) * ntiles + i / tile_size] + rank[i];
  pairs[pos] = make_uint2(c & ((1u << lo_bits) - 1u), (unsigned int)i);
}

constexpr int BIN_TPB = 256;
__global__ __launch_bounds__(BIN_TPB) void bin_sort_kernel(const uint2 *__restrict__ pairs,
                                                           const unsigned int *__restrict__ scanned, int ntiles,
                                                           int nb_lo, int *__restrict__ order) {
  extern __shared__ unsigned int lds_lo[];  // nb_lo counters, then BIN_TPB partial sums
  unsigned int *part = lds_lo + nb_lo;
  const unsigned int start = scanned[(size_t)blockIdx.x * ntiles], end = scanned[(size_t)(blockIdx.x + 1) * ntiles];
  if (start == end) return;  // block-uniform
  for (int b = threadIdx.x; b < nb_lo; b += BIN_TPB) lds_lo[b] = 0u;
  __syncthreads();
  for (unsigned int i = start + threadIdx.x; i < end; i += BIN_TPB) atomicAdd(&lds_lo[pairs[i].x], 1u);
  __syncthreads();
  // exclusive scan of the nb_lo counters: per-thread runs, Hillis-Steele over the BIN_TPB run totals
  const int per = (nb_lo + BIN_TPB - 1) / BIN_TPB;
  const int b0 = threadIdx.x * per;
  unsigned int run = 0u;
  for (int b = b0; b < b0 + per && b < nb_lo; ++b) run += lds_lo[b];
  part[threadIdx.x] = run;
  __syncthreads();
  for (int o = 1; o < BIN_TPB; o <<= 1) {
    const unsigned int add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned int base = part[threadIdx.x] - run;  // exclusive prefix of this thread's run
  for (int b = b0; b < b0 + per && b < nb_lo; ++b) {
    const unsigned int c = lds_lo[b];
    lds_lo[b] = base;
    base += c;
  }
  __syncthreads();
  for (unsigned int i = start + threadIdx.x; i < end; i += BIN_TPB) {
    const uint2 pr = pairs[i];
    order[start + atomicAdd(&lds_lo[pr.x], 1u)] = (int)pr.y;
  }
}

}  // namespace


// in-place exclusive sum of the workspace's first `count` histogram counters.  The temporary storage hipcub asks for
// is checked against what the workspace holds (a size query allocates nothing): the static scratch of
// pclmi_knn_dev_noalloc reserves a fixed slice for it, and an under-estimate would write past the device array.
static int scan_hist(Workspace *w, int count, hipStream_t s) {
  size_t need = 0;
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, w->hist, w->hist, count, s));
  PCLMI_CHECK(need <= w->cap_tmp, w->fixed ? PCLMI_ERR_UNSUPPORTED : PCLMI_ERR_HIP,
              "the scan of %d counters needs %zu bytes of temporary storage, the workspace holds %zu", count, need,
              w->cap_tmp);
  size_t tmp = w->cap_tmp;
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(w->sort_tmp, tmp, w->hist, w->hist, count, s));
  return PCLMI_OK;
}

int launch_morton_order(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq,
                        size_t q_stride_f, const float *bb_lo, const float *bb_hi, hipStream_t s,
                        const int32_t **d_order, bool stable) {
  if (!bb_lo) bb_lo = t->bb_lo;
  if (!bb_hi) bb_hi = t->bb_hi;
  *d_order = nullptr;
  if (nq <= 64) return PCLMI_OK;  // a single wave: order is irrelevant
  // cells: one query per cell or fewer (measured on the 1M-point surface cloud, k = 16 normals: 4 queries per cell
  // 0.597 ms, 0.5 per cell 0.565 ms, 0.06 per cell 0.578 ms; on the uniform cloud the three are equal), 2..8 bits per axis
  int lg = 0;
  while (((size_t)1 << lg) < nq) ++lg;
  int bits = ((lg + 1) / 3) * 3;
  if (w->fixed) bits = ((lg - 2) / 3) * 3;  // the allocation-free entry point's histogram space is sized for 4 per cell
  bits = bits < 6 ? 6 : (bits > 24 ? 24 : bits);
  if (const char *e = getenv("PCLMI_CELL_BITS")) bits = atoi(e);  // development override
  const size_t nbins = (size_t)1 << bits;
  // large batches: the two-level counting sort with LDS counters (see tile_count_kernel)
  static const bool atomic_order = [] {
    const char *e = getenv("PCLMI_ORDER");  // development switch: PCLMI_ORDER=atomic keeps the one-level sort
    return e && e[0] == 'a';
  }();
  const bool two_level = !stable && !atomic_order && nq >= 32768;
  int lo_bits = bits / 2, hi_bits = bits - lo_bits;
  int nb_hi = 1 << hi_bits, nb_lo = 1 << lo_bits;
  // tube order (see tube_key) when the indexed cloud fills its box; PCLMI_ORDER=tube / hilbert forces one
  static const int order_env = [] {
    const char *e = getenv("PCLMI_ORDER");
    return !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'h' ? 2 : 0));
  }();
  int gy = 0, gz = 0, tube_p = 0;
  float tube_sc[3] = {0.f, 0.f, 0.f};
  if (two_level && !w->fixed && order_env != 2 && (order_env == 1 || t->fill >= 0.2f)) {
    const double ex = (double)bb_hi[0] - bb_lo[0], ey = (double)bb_hi[1] - bb_lo[1], ez = (double)bb_hi[2] - bb_lo[2];
    if (ex > 0.0 && ey > 0.0 && ez > 0.0) {
      static const double tube_scale = [] {
        const char *e = getenv("PCLMI_TUBE_SCALE");  // development knob
        return e ? atof(e) : 1.0;
      }();
      const double side = tube_scale * cbrt(64.0 * ex * ey * ez / (double)nq);  // of a cube with one wavefront's queries
      gy = (int)(ey / side + 0.5);
      gz = (int)(ez / side + 0.5);
      gy = gy < 1 ? 1 : gy;
      gz = gz < 1 ? 1 : gz;
      while (gy > 64 || gz > 64) {  // counters of one tile live in LDS (64 x 64 tubes at most)
        gy = gy > 1 ? (gy + 1) / 2 : 1;
        gz = gz > 1 ? (gz + 1) / 2 : 1;
      }
      static const bool tube_rows = [] {
        const char *e = getenv("PCLMI_TUBE_WALK");  // development switch: rows = boustrophedon rows of tubes
        return e && e[0] == 'r';
      }();
      if (!tube_rows) {
        tube_p = 1;
        while (tube_p < gy || tube_p < gz) tube_p <<= 1;
      }
      tube_sc[0] = (float)(1024.0 / ex);
      tube_sc[1] = (float)(gy / ey);
      tube_sc[2] = (float)(gz / ez);
      lo_bits = 10;
      nb_lo = 1024;
      nb_hi = tube_p ? tube_p * tube_p : gy * gz;
    }
  }
  static const int tile_env = [] {
    const char *e = getenv("PCLMI_TILE");  // development knob: queries per tile of the two-level sort
    return e ? atoi(e) : 4096;
  }();
  const int tile_q = tile_env >= 1024 ? tile_env : 4096;
  int ntiles = (int)((nq + tile_q - 1) / tile_q);
  ntiles = ntiles > 512 ? 512 : ntiles;
  const int tile_size = (int)((nq + ntiles - 1) / ntiles);
  int rc = w->ensure_queries(nq, stable ? 1 : two_level ? (size_t)nb_hi * ntiles + 1 : nbins);
  if (rc) return rc;
  if (two_level && (rc = w->ensure_scratch(nq * sizeof(uint2)))) return rc;
  float sc[3];
  for (int d = 0; d < 3; ++d) {
    float ext = bb_hi[d] - bb_lo[d];
    sc[d] = ext > 0.f ? 1024.f / ext : 0.f;
  }
  TimingScope ts(PCLMI_TIMER_ORDER, s);
  const unsigned grid = (unsigned)((nq + 255) / 256);
  if (two_level) {
    hipLaunchKernelGGL(tile_count_kernel, dim3(ntiles), dim3(TILE_TPB),
                       ((size_t)nb_hi + (size_t)tube_p * tube_p) * sizeof(unsigned int), s, d_q,
                       (int)q_stride_f, (int)nq, bb_lo[0], bb_lo[1], bb_lo[2], gy ? tube_sc[0] : sc[0],
                       gy ? tube_sc[1] : sc[1], gy ? tube_sc[2] : sc[2], 30 - bits, lo_bits, nb_hi, tile_size, ntiles, gy, gz,
                       tube_p, w->keys_in, w->ranks, w->hist);
    RC(scan_hist(w, nb_hi * ntiles + 1, s));
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(grid), dim3(256), 0, s, w->keys_in, w->ranks, (int)nq, lo_bits, tile_size,
                       ntiles, w->hist, (uint2 *)w->scratch);
    hipLaunchKernelGGL(bin_sort_kernel, dim3(nb_hi), dim3(BIN_TPB), (size_t)(nb_lo + BIN_TPB) * sizeof(unsigned int), s,
                       (const uint2 *)w->scratch, w->hist, ntiles, nb_lo, w->vals_out);
    PCLMI_HIP(hipGetLastError());
    *d_order = w->vals_out;
    return PCLMI_OK;
  }
  if (stable) {
    // Reproducible order (ICP: the order fixes the summation order of the fp64 moments): a stable radix
    // sort of (cell, row) pairs instead of the atomic ranks, whose order inside a cell varies from run to run.
    // All 30 bits of the curve index: the sort is set-up work, and the per-lane ICP search gains from every
    // level of locality (1M x 1M surface cloud: 0.121 ms per iteration with 18 bits, 0.109 ms with 24 or 30).
    if (!getenv("PCLMI_CELL_BITS")) bits = 30;
    hipLaunchKernelGGL(cell_only_kernel, dim3(grid), dim3(256), 0, s, d_q, (int)q_stride_f, (int)nq, bb_lo[0], bb_lo[1],
                       bb_lo[2], sc[0], sc[1], sc[2], 30 - bits, w->keys_in);
    launch_iota((int *)w->ranks, nq, s);
    size_t need = 0;
    PCLMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, (const unsigned int *)w->keys_in, (unsigned int *)nullptr,
                                                 (const int *)w->ranks, (int *)w->vals_out, (int)nq, 0, bits, s));
    const size_t keys_bytes = (nq * sizeof(unsigned int) + 255) & ~(size_t)255;
    rc = w->ensure_scratch(keys_bytes + need);
    if (rc) return rc;
    PCLMI_HIP(hipcub::DeviceRadixSort::SortPairs((char *)w->scratch + keys_bytes, need, (const unsigned int *)w->keys_in,
                                                 (unsigned int *)w->scratch, (const int *)w->ranks, (int *)w->vals_out,
                                                 (int)nq, 0, bits, s));
    *d_order = w->vals_out;
    return PCLMI_OK;
  }
  PCLMI_HIP(hipMemsetAsync(w->hist, 0, nbins * sizeof(unsigned int), s));
  hipLaunchKernelGGL(cell_count_kernel, dim3(grid), dim3(256), 0, s, d_q, (int)q_stride_f, (int)nq,
                     bb_lo[0], bb_lo[1], bb_lo[2], sc[0], sc[1], sc[2], 30 - bits, w->keys_in, w->ranks, w->hist);
  RC(scan_hist(w, (int)nbins, s));
  hipLaunchKernelGGL(cell_scatter_kernel, dim3(grid), dim3(256), 0, s, w->keys_in, w->ranks, (int)nq, w->hist,
                     w->vals_out);
  PCLMI_HIP(hipGetLastError());
  *d_order = w->vals_out;
  return PCLMI_OK;
}

// vp != nullptr: the fused surface-normal variant (d_d2 receives nq x 4 floats, d_idx is unused)
// (Measured and dropped: the pre-pass as a kernel of its own at three times the occupancy, lists handed over through
// HBM - 100 us for 1M queries on its own against the 40 us it adds to the fused kernel's 605.)
template <int K, int C, int WPB>
static int launch_union_ix(const pclmi_tree *t, Workspace *, const float *d_q, size_t nq, size_t qsf,
                           const int32_t *order, int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s,
                           const float *vp = nullptr) {
  // a sparse batch that leaves wavefront slots empty is spread over them, down to one query per wavefront (whose
  // "union" is then that query's own ball, walked with scalar loads), see load_lane_query: K = 64 has no per-lane
  // twin, and below that knn_sparse sends batches of up to 4096 queries here
  const int qpw = (nq * 6 < t->n) ? lane_queries_per_wave(nq, K == 64 ? 1024 : K == 32 ? 2048 : 4096) : 64;
  unsigned waves = (unsigned)((nq + qpw - 1) / qpw);
  unsigned blocks = (waves + WPB - 1) / WPB;
  blocks = (blocks + 7) & ~7u;
  if (vp)
    hipLaunchKernelGGL((knn_union_ix_kernel<K, C, WPB, true>), dim3(blocks), dim3(64 * WPB), 0, s, t->view(), d_q,
                       (int)qsf, order, (int)nq, k_eff, d_idx, d_d2, vp[0], vp[1], vp[2], qpw, k_eff);
  else
    hipLaunchKernelGGL((knn_union_ix_kernel<K, C, WPB, false>), dim3(blocks), dim3(64 * WPB), 0, s, t->view(), d_q,
                       (int)qsf, order, (int)nq, k_eff, d_idx, d_d2, 0.f, 0.f, 0.f, qpw, k_eff);
  return PCLMI_OK;
}

// k in 65 .. kTiledMaxK: the list in tiles of 64, one launch of the K = 64 instance per tile - the first as for
// k = 64, every further one restricted to keys above the last key of the tile before it (LOW).  The reference's own
// differential test searches with k = 512 (test/test_search.cpp:624-631); the per-lane heap kernel this replaces did
// 2e7 queries/s at k = 128.
static int launch_union_tiled(const pclmi_tree *t, const float *d_q, size_t nq, size_t qsf, const int32_t *order,
                              int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s) {
  // the first tile spreads a sparse batch over the wavefront slots like k = 64 does; the further tiles run without the
  // pre-pass, where lanes without a query would outvote the others in the walk's nearest-first order: full wavefronts
  const int qpw0 = (nq * 6 < t->n) ? lane_queries_per_wave(nq, 1024) : 64;
  for (int col = 0; col < k_eff; col += 64) {
    const int cols = std::min(64, k_eff - col);
    const int qpw = col == 0 ? qpw0 : 64;
    const unsigned waves = (unsigned)((nq + qpw - 1) / qpw);
    const unsigned blocks = (waves + 7) & ~7u;
    if (col == 0)
      hipLaunchKernelGGL((knn_union_ix_kernel<64, 128, 1, false, false>), dim3(blocks), dim3(64), 0, s, t->view(), d_q,
                         (int)qsf, order, (int)nq, cols, d_idx, d_d2, 0.f, 0.f, 0.f, qpw, k_eff);
    else
      hipLaunchKernelGGL((knn_union_ix_kernel<64, 128, 1, false, true>), dim3(blocks), dim3(64), 0, s, t->view(), d_q,
                         (int)qsf, order, (int)nq, cols, d_idx + col, d_d2 + col, 0.f, 0.f, 0.f, qpw, k_eff);
  }
  return PCLMI_OK;
}

template <int K, int C>
static void launch_lane_ix(const pclmi_tree *t, const float *d_q, size_t nq, size_t qsf, const int32_t *order,
                           int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s, const float *vp = nullptr) {
  const int levels = lane_nearest_levels(t->M);
  const int qpw = lane_queries_per_wave(nq, K <= 16 ? 2048 : 1024);
  const unsigned blocks = (((unsigned)((nq + qpw - 1) / qpw)) + 7) & ~7u;
  const size_t lds = (size_t)levels * 64 * sizeof(float);
  if (vp)
    hipLaunchKernelGGL((knn_lane_ix_kernel<K, C, true>), dim3(blocks), dim3(64), lds, s, t->view(), d_q, (int)qsf, order,
                       (int)nq, levels, k_eff, d_idx, d_d2, vp[0], vp[1], vp[2], qpw);
  else
    hipLaunchKernelGGL((knn_lane_ix_kernel<K, C, false>), dim3(blocks), dim3(64), lds, s, t->view(), d_q, (int)qsf, order,
                       (int)nq, levels, k_eff, d_idx, d_d2, 0.f, 0.f, 0.f, qpw);
}

// sparse batch: the per-lane kernels (measured crossover); PCLMI_KNN=union / lane forces one
static bool knn_sparse(const pclmi_tree *t, size_t nq) {
  static const int forced = [] {
    const char *e = getenv("PCLMI_KNN");
    return !e ? 0 : (e[0] == 'u' ? 1 : (e[0] == 'l' ? 2 : 0));
  }();
  static const size_t ratio = [] {
    const char *e = getenv("PCLMI_KNN_SPARSE_RATIO");
    return e ? (size_t)atoi(e) : (size_t)6;  // 1M-point target, k = 16: 100k queries lane 0.237 / union 0.282 ms, 250k 0.410 / 0.288
  }();
  // up to 4096 queries the union kernel takes sparse batches too: launch_union_ix gives every query a wavefront of
  // its own, whose walk is then that one query's, on scalar loads (k = 16, 1M-point target: 1 query 26.8 us against
  // the per-lane kernel's 31.0, 64 queries 38 / 48, 1024 51 / 60, 4096 75 / 90; 16 384 - four queries per wavefront -
  // 126 / 119)
  return forced ? forced == 2 : (nq > 4096 && nq * ratio < t->n);
}

#define PCLMI_STR2(x) #x
#define PCLMI_STR(x) PCLMI_STR2(x)
// the kernel launch_knn picks for a batch (the same rules, for reports: bench.py names the kernel it timed)
constexpr int kTiledMaxK = 1024;
static bool knn_force_heap();
// k > 64 in tiles of 64 (launch_union_tiled) or on the per-lane heap kernel: the tiles' further launches walk for whole
// wavefronts of 64 queries without a pre-pass, which loses on small batches, whose 64 queries are far apart (1M-point
// target, tiles / heap: k = 128 - 4000 queries 6.5 / 4.5 ms, 10k 4.2 / 4.5, 30k 2.7 / 5.6, 1M 8.7 / 49.9;
// k = 512 - 10k 28 / 17, 30k 16.8 / 22.2, 100k 24 / 37)
static bool knn_tiled(size_t nq, int k_eff) { return k_eff > 64 && k_eff <= kTiledMaxK && nq >= 16384 && !knn_force_heap(); }
const char *knn_kernel_name(const pclmi_tree *t, size_t nq, int k_eff) {
  if (k_eff <= 1) return "knn1_union_kernel<4> / knn1_lane_kernel<4>";
  if (k_eff <= 32 && knn_sparse(t, nq))
    return k_eff <= 4 ? "knn_lane_ix_kernel<4,32>" : k_eff <= 8 ? "knn_lane_ix_kernel<8,32>"
           : k_eff <= 16 ? "knn_lane_ix_kernel<16,40>" : "knn_lane_ix_kernel<32,64>";
  if (k_eff <= 2) return "knn_union_ix_kernel<2,16,1>";
  if (k_eff <= 4) return "knn_union_ix_kernel<4,16,1>";
  if (k_eff <= 8) return "knn_union_ix_kernel<8," PCLMI_STR(PCLMI_IX8) ",1>";
  if (k_eff <= 16) return "knn_union_ix_kernel<16," PCLMI_STR(PCLMI_IX16) ",1>";
  if (k_eff <= 32) return "knn_union_ix_kernel<32," PCLMI_STR(PCLMI_IX32) ",1>";
  if (k_eff <= 64) return "knn_union_ix_kernel<64,128,1>";
  if (knn_tiled(nq, k_eff)) return "knn_union_ix_kernel<64,128,1> x tiles of 64";
  return "knn_heap_kernel";
}

bool knn_normals_fused(int k_eff) { return k_eff >= 2 && k_eff <= 32; }

// the fused search + plane fit (normals_finish) for 2 <= k_eff <= 32: d_out receives nq x (nx, ny, nz, curvature)
int launch_knn_normals(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t qsf, const int32_t *order,
                       int k_eff, const float vp[3], float *d_out, hipStream_t s) {
  if (nq == 0) return PCLMI_OK;
  PCLMI_CHECK(knn_normals_fused(k_eff), PCLMI_ERR_INVALID, "fused normals need 2 <= k <= 32 (got %d)", k_eff);
  PCLMI_CHECK(nq < (size_t)1 << 31, PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 queries in one call");
  TimingScope ts(PCLMI_TIMER_KNN, s);
  int rc = PCLMI_OK;
  if (knn_sparse(t, nq)) {
    if (k_eff <= 4) launch_lane_ix<4, 32>(t, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
    else if (k_eff <= 8) launch_lane_ix<8, 32>(t, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
    else if (k_eff <= 16) launch_lane_ix<16, 40>(t, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
    else launch_lane_ix<32, 64>(t, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  } else if (k_eff <= 2) {
    rc = launch_union_ix<2, 16, 1>(t, w, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  } else if (k_eff <= 4) {
    rc = launch_union_ix<4, 16, 1>(t, w, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  } else if (k_eff <= 8) {
    rc = launch_union_ix<8, PCLMI_IX8, 1>(t, w, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  } else if (k_eff <= 16) {
    rc = launch_union_ix<16, PCLMI_IX16, 1>(t, w, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  } else {
    rc = launch_union_ix<32, PCLMI_IX32, 1>(t, w, d_q, nq, qsf, order, k_eff, nullptr, d_out, s, vp);
  }
  if (rc) return rc;
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

// development switch PCLMI_BIGK=heap: k > 64 on the per-lane heap kernel (A/B against the tiles)
static bool knn_force_heap() {
  static const bool f = [] {
    const char *e = getenv("PCLMI_BIGK");
    return e && e[0] == 'h';
  }();
  return f;
}

int launch_knn(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t qsf,
               const int32_t *order, int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s) {
  if (nq == 0 || k_eff <= 0) return PCLMI_OK;
  PCLMI_CHECK(nq < (size_t)1 << 31, PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 queries in one call");
  TimingScope ts(PCLMI_TIMER_KNN, s);
  int rc = PCLMI_OK;
  if (k_eff == 1) {
    constexpr int WPB = 4;
    unsigned waves = (unsigned)((nq + 63) / 64);
    unsigned blocks = ((waves + WPB - 1) / WPB + 7) & ~7u;
    // Per-lane walk or wave-uniform walk: the wave-uniform one wins when the batch is about as dense as the target
    // or denser (its 64 queries then share most of their nodes: 4.9e9 against 3.5e9 queries/s with 3 queries per
    // target point, 0.254 against 0.281 ms at 1M x 1M), the per-lane one on sparse batches (0.11 against 0.16 ms for
    // 100k queries on a 1M-point target).  PCLMI_NN1=union / lane forces one of them (development switch).
    static const int forced = [] {
      const char *e = getenv("PCLMI_NN1");
      return !e ? 0 : (e[0] == 'u' ? 1 : (e[0] == 'l' ? 2 : 0));
    }();
    // Batches of 4..4096 queries take the wave-uniform walk even when sparse: spread one query per wavefront (below),
    // it walks that query's ball on scalar loads (64 queries 21.7 against 25.8 us, 1024 26.4 / 32.2, 4096 36.6 / 40.1;
    // a single query 11.6 / 10.4, 16 384 queries 54.9 / 50.2).
    const bool per_lane = forced ? forced == 2 : (nq * 2 < t->n && (nq > 4096 || nq < 4));
    if (per_lane) {
      const int levels = lane_nearest_levels(t->M);
      const int qpw = lane_queries_per_wave(nq, 2048);
      waves = (unsigned)((nq + qpw - 1) / qpw);
      blocks = ((waves + WPB - 1) / WPB + 7) & ~7u;
      hipLaunchKernelGGL((knn1_lane_kernel<WPB>), dim3(blocks), dim3(64 * WPB), (size_t)WPB * levels * 64 * sizeof(float),
                         s, t->view(), d_q, (int)qsf, order, (int)nq, levels, d_idx, d_d2, qpw);
    } else {
      const int qpw = (nq * 2 < t->n) ? lane_queries_per_wave(nq, 4096) : 64;  // sparse here: forced, or a small batch
      waves = (unsigned)((nq + qpw - 1) / qpw);
      blocks = ((waves + WPB - 1) / WPB + 7) & ~7u;
      hipLaunchKernelGGL((knn1_union_kernel<WPB>), dim3(blocks), dim3(64 * WPB), 0, s, t->view(), d_q, (int)qsf,
                         order, (int)nq, d_idx, d_d2, qpw);
    }
  } else if (k_eff <= 32 && knn_sparse(t, nq)) {
    if (k_eff <= 4) launch_lane_ix<4, 32>(t, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
    else if (k_eff <= 8) launch_lane_ix<8, 32>(t, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
    else if (k_eff <= 16) launch_lane_ix<16, 40>(t, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
    else launch_lane_ix<32, 64>(t, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 2) {
    rc = launch_union_ix<2, 16, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 4) {
    rc = launch_union_ix<4, 16, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 8) {
    rc = launch_union_ix<8, PCLMI_IX8, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 16) {
    rc = launch_union_ix<16, PCLMI_IX16, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 32) {
    rc = launch_union_ix<32, PCLMI_IX32, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (k_eff <= 64) {
    // the reference's own tests search with k = 64 (test/test_search.cpp:624-631): same kernel, one wave per SIMD.
    // Sparse batches too: there is no per-lane instance at this K, and against the heap kernel the union walk wins
    // at every batch size (1M-point target: 1000 queries 2.0 vs 2.4 ms, 10k 1.25 vs 2.6, 100k 0.93 vs 4.4)
    rc = launch_union_ix<64, 128, 1>(t, w, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else if (knn_tiled(nq, k_eff)) {
    rc = launch_union_tiled(t, d_q, nq, qsf, order, k_eff, d_idx, d_d2, s);
  } else {
    size_t bytes = (size_t)k_eff * nq * sizeof(unsigned long long);
    rc = w->ensure_scratch(bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(knn_heap_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, s, t->view(), d_q,
                       (int)qsf, order, (int)nq, k_eff, (unsigned long long *)w->scratch, d_idx, d_d2);
  }
  if (rc) return rc;
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

}  // namespace pclmi
