// radius_kernels.hip - fixed-radius neighbour search on gfx950 (MI355X), wave64.
//
// Replaces the FLANN traversal behind pcl::KdTreeFLANN::radiusSearch
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:115-157): all points with d2 < (float)(r*r)
// (strict), optionally only the nearest max_nn, ascending by (d2, index) when sorted, by index
// otherwise.  The output is variable length, so the search runs as
//   1. count pass   - the wave-uniform union traversal of traverse.hpp, every lane counting the hits
//                     of its own query AND (without max_nn) keeping its first 64 keys
//                     (d2 bits << 32 | index, or index << 32 | d2 bits for unsorted output) in a
//                     fixed-pitch provisional row;
//   2. exclusive scan of the counts -> CSR offsets (the only host synchronisation: the total);
//   3a. no query with more than 64 hits (the common case): the provisional rows are sorted straight into
//                     the CSR idx / d2 arrays (one key per lane, bitonic network with DPP exchanges) - done;
//   3b. otherwise: fill pass - the same traversal again, every lane appending its keys to its own CSR
//                     segment; per-segment sort (up to 64 entries as in 3a; up to 1024: bitonic network in
//                     LDS; hipcub::DeviceSegmentedSort for the rare longer ones); split of the keys into the
//                     idx / d2 arrays (and truncation to max_nn).
// Queries are visited in Morton-cell order like the kNN path; segments are laid out in the caller's
// query order.
#include <cstring>

#include <hipcub/hipcub.hpp>

#include "eig3.hpp"
#include "pclmi_internal.h"
#include "traverse.hpp"

namespace pclmi {
namespace {

using namespace dev;
typedef unsigned long long u64;

constexpr int R_WPB = 4;

// MODE 0: count only.  MODE 1: fill - append keys to the CSR segment of the query (offsets known).
// MODE 2: count AND write the first PROV_CAP keys of every query to its fixed-pitch provisional row
// keys[qi * PROV_CAP ..]; *n_over counts the queries with more hits than that.  When none overflows (the common
// case: ~32 neighbours), the rows are sorted straight into the CSR output and the second traversal is skipped.
constexpr int PROV_CAP = 64;
template <int MODE>
__global__ __launch_bounds__(64 * R_WPB) void radius_union_kernel(TreeView t, const float *__restrict__ q,
                                                                   int q_stride_f,
                                                                   const int *__restrict__ order, int nq,
                                                                   float r2, float tau_r, int by_index,
                                                                   unsigned int *__restrict__ counts,
                                                                   const u64 *__restrict__ offsets,
                                                                   u64 *__restrict__ keys,
                                                                   unsigned int *__restrict__ n_over) {
  constexpr bool FILL = MODE == 1;
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<R_WPB>(q, q_stride_f, order, nq, lq)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  // a box can hold a hit iff its distance is < r2, i.e. <= tau_r = the largest float below r2;
  // lanes without a searchable query get a negative bound and never steer the wave
  const float tau = lq.searchable ? tau_r : -1.f;
  unsigned int cnt = 0;
  u64 *out = nullptr;
  if (FILL && lq.alive) out = keys + offsets[lq.qi];
  if (MODE == 2 && lq.alive) out = keys + (size_t)lq.qi * PROV_CAP;

  LaneStack st{0, 0, lane};
  st.push(1);
  const cgrp_ptr box2 = (cgrp_ptr)t.box;
  const cgrp_ptr leaf = (cgrp_ptr)t.leaf;
  const int M = t.M;
  while (st.sp > 0) {
    const int node = st.pop();
    if (node >= M) {
      // no box test here: the bound of a radius search never moves, so expand_inner's test when it pushed the bucket
      // is the whole truth (a tree of one bucket is scanned untested: 16 exact candidate tests)
      const cgrp_ptr rec = leaf + (size_t)(node - M) * G;
      f16v grp[G];
#pragma unroll
      for (int g = 0; g < G; ++g) grp[g] = rec[g];
      const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
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
          // padded slots sit at 3e38: their d2 is +inf and never < r2; NaN queries never pass
          if (dd[j] < r2) {
            if (FILL || (MODE == 2 && cnt < (unsigned int)PROV_CAP)) {
              const unsigned int ib = __float_as_uint(grp[g][12 + j]), db = __float_as_uint(dd[j]);
              out[cnt] = by_index ? (((u64)ib << 32) | db) : (((u64)db << 32) | ib);
            }
            ++cnt;
          }
        }
      }
    } else {
      expand_inner(st, box2, node, M, qx, qy, qz, tau);
    }
  }
  if (!FILL && lq.alive) counts[lq.qi] = cnt;
  if (MODE == 2) {
    const int over = __popcll(__ballot(cnt > (unsigned int)PROV_CAP));
    if (over && lane == 0) atomicAdd(n_over, (unsigned int)over);
  }
}

// Sparse batches (fewer than a quarter as many queries as target points): every lane walks for itself
// (traverse.hpp lane_traverse) under the fixed bound - the union of 64 far-apart search balls is most of a subtree
// (10 000 queries on a 1M-point target: 0.67 ms for the union walk).  Same three modes and outputs.
template <int MODE>
__global__ __launch_bounds__(64) void radius_lane_kernel(TreeView t, const float *__restrict__ q, int q_stride_f,
                                                         const int *__restrict__ order, int nq, int levels, float r2,
                                                         float tau_r, int by_index, unsigned int *__restrict__ counts,
                                                         const u64 *__restrict__ offsets, u64 *__restrict__ keys,
                                                         unsigned int *__restrict__ n_over, int qpw) {
  constexpr bool FILL = MODE == 1;
  extern __shared__ float far_lds[];  // levels x 64 floats
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<1>(q, q_stride_f, order, nq, lq, qpw)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  unsigned int cnt = 0;
  u64 *out = nullptr;
  if (FILL && lq.alive) out = keys + offsets[lq.qi];
  if (MODE == 2 && lq.alive) out = keys + (size_t)lq.qi * PROV_CAP;
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  lane_traverse(
      t, qx, qy, qz, lq.searchable, tau_r, reinterpret_cast<float(*)[64]>(far_lds), lane, 1,
      [&](int leafno, float tau) {
        const float4 *rec = reinterpret_cast<const float4 *>(t.leaf + (size_t)leafno * (4 * L));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 X = rec[g * 4 + 0], Y = rec[g * 4 + 1], Z = rec[g * 4 + 2], I = rec[g * 4 + 3];
          const f2v d01 = sq3_pk(q2x - f2v{X.x, X.y}, q2y - f2v{Y.x, Y.y}, q2z - f2v{Z.x, Z.y});
          const f2v d23 = sq3_pk(q2x - f2v{X.z, X.w}, q2y - f2v{Y.z, Y.w}, q2z - f2v{Z.z, Z.w});
          const float dd[4] = {d01.x, d01.y, d23.x, d23.y};
          const float ii[4] = {I.x, I.y, I.z, I.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (dd[j] < r2) {  // padded slots are at +inf
              if (FILL || (MODE == 2 && cnt < (unsigned int)PROV_CAP)) {
                const unsigned int ib = __float_as_uint(ii[j]), db = __float_as_uint(dd[j]);
                out[cnt] = by_index ? (((u64)ib << 32) | db) : (((u64)db << 32) | ib);
              }
              ++cnt;
            }
          }
        }
        return tau;
      },
      [] {});
  if (!FILL && lq.alive) counts[lq.qi] = cnt;
  if (MODE == 2) {
    const int over = __popcll(__ballot(cnt > (unsigned int)PROV_CAP));
    if (over && lane == 0) atomicAdd(n_over, (unsigned int)over);
  }
}

// ---- fused radius search + surface normal ----------------------------------------------------------
// NormalEstimation with setRadiusSearch (features/impl/normal_3d.hpp:46-98 over Feature's radius binding,
// features/impl/feature.hpp:143-159): the plane fit needs the SUM of the neighbours' moments, not the list, so the
// variable-length output of the radius search (count pass, scan, fill / sort, CSR arrays) never has to exist.  Every
// hit's bucket slot goes into a small per-lane LDS buffer; when a bucket's worth of room is no longer free the lanes
// drain their own entries - gather the slot row (touched a moment ago), add the fp64 moments about the query point.
// Draining per lane (not per candidate under the wave's union mask) keeps the fp64 work proportional to a lane's own
// hits.  The epilogue is the one of the k-NN variant (knn_kernels.hip normals_finish): smallest eigenpair,
// curvature, flip towards the viewpoint, 16 bytes per query.
constexpr int RN_C = 32;  // buffer entries per lane (>= 2 buckets)
struct MomentAcc {
  double n = 0, sx = 0, sy = 0, sz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
  __device__ __forceinline__ void drain(unsigned int (*buf)[64], int lane, int &cnt, const float4 *__restrict__ rows,
                                        float qx, float qy, float qz) {
    int mx = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    for (int e = 0; e < mx; e += 8) {
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        p[u] = rows[(e + u < cnt) ? buf[min(e + u, RN_C - 1)][lane] : 0u];  // safe index: the eight gathers fly together
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (e + u < cnt) {
          const double dx = (double)p[u].x - (double)qx, dy = (double)p[u].y - (double)qy, dz = (double)p[u].z - (double)qz;
          n += 1.0;
          sx += dx; sy += dy; sz += dz;
          xx += dx * dx; xy += dx * dy; xz += dx * dz;
          yy += dy * dy; yz += dy * dz; zz += dz * dz;
        }
      }
    }
    cnt = 0;
  }
  __device__ __forceinline__ void finish(const LaneQuery &lq, float vpx, float vpy, float vpz, float4 *__restrict__ out) {
    if (!lq.alive) return;
    const float nanf_ = __builtin_nanf("");
    float4 res = make_float4(nanf_, nanf_, nanf_, nanf_);
    if (lq.searchable && n > 0.0) {
      const double s1[3] = {sx, sy, sz}, s2[6] = {xx, xy, xz, yy, yz, zz}, o[3] = {lq.x, lq.y, lq.z};
      double plane[4], curv;
      eig::plane_from_moments(n, s1, s2, o, plane, curv);
      const double c = ((double)vpx - lq.x) * plane[0] + ((double)vpy - lq.y) * plane[1] + ((double)vpz - lq.z) * plane[2];
      const double sgn = c < 0.0 ? -1.0 : 1.0;
      res = make_float4((float)(sgn * plane[0]), (float)(sgn * plane[1]), (float)(sgn * plane[2]), (float)curv);
    }
    out[lq.qi] = res;
  }
};

constexpr int RN_WPB = 2;
__global__ __launch_bounds__(64 * RN_WPB) void radius_normals_union_kernel(TreeView t, const float *__restrict__ q,
                                                                           int q_stride_f,
                                                                           const int *__restrict__ order, int nq,
                                                                           float r2, float tau_r, float vpx, float vpy,
                                                                           float vpz, float4 *__restrict__ out) {
  __shared__ unsigned int lds[RN_WPB][RN_C][64];
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<RN_WPB>(q, q_stride_f, order, nq, lq)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  const float tau = lq.searchable ? tau_r : -1.f;
  unsigned int(*buf)[64] = lds[threadIdx.x >> 6];
  const float4 *rows = reinterpret_cast<const float4 *>(t.slot);
  int cnt = 0;
  MomentAcc acc;
  LaneStack st{0, 0, lane};
  st.push(1);
  const cgrp_ptr box2 = (cgrp_ptr)t.box;
  const cgrp_ptr leaf = (cgrp_ptr)t.leaf;
  const int M = t.M;
  while (st.sp > 0) {
    const int node = st.pop();
    if (node >= M) {
      const cgrp_ptr rec = leaf + (size_t)(node - M) * G;  // (no second box test: the bound never moves)
      f16v grp[G];
#pragma unroll
      for (int g = 0; g < G; ++g) grp[g] = rec[g];
      if (__any(cnt > RN_C - L)) acc.drain(buf, lane, cnt, rows, qx, qy, qz);
      const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
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
          if (dd[j] < r2) {  // strict (kdtree_flann.hpp:115-157); padded slots are at +inf, NaN queries never pass
            buf[cnt][lane] = (unsigned int)(node - M) * L + g * 4 + j;
            ++cnt;
          }
        }
      }
    } else {
      expand_inner(st, box2, node, M, qx, qy, qz, tau);
    }
  }
  acc.drain(buf, lane, cnt, rows, qx, qy, qz);
  acc.finish(lq, vpx, vpy, vpz, out);
}

__global__ __launch_bounds__(64) void radius_normals_lane_kernel(TreeView t, const float *__restrict__ q, int q_stride_f,
                                                                 const int *__restrict__ order, int nq, int levels,
                                                                 float r2, float tau_r, float vpx, float vpy, float vpz,
                                                                 float4 *__restrict__ out, int qpw) {
  extern __shared__ float far_lds[];  // levels x 64 floats
  __shared__ unsigned int buf[RN_C][64];
  const int lane = threadIdx.x & 63;
  LaneQuery lq;
  if (!load_lane_query<1>(q, q_stride_f, order, nq, lq, qpw)) return;
  const float qx = lq.x, qy = lq.y, qz = lq.z;
  const float4 *rows = reinterpret_cast<const float4 *>(t.slot);
  int cnt = 0;
  MomentAcc acc;
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  lane_traverse(
      t, qx, qy, qz, lq.searchable, tau_r, reinterpret_cast<float(*)[64]>(far_lds), lane, 1,
      [&](int leafno, float tau) {
        const float4 *rec = reinterpret_cast<const float4 *>(t.leaf + (size_t)leafno * (4 * L));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 X = rec[g * 4 + 0], Y = rec[g * 4 + 1], Z = rec[g * 4 + 2];
          const f2v d01 = sq3_pk(q2x - f2v{X.x, X.y}, q2y - f2v{Y.x, Y.y}, q2z - f2v{Z.x, Z.y});
          const f2v d23 = sq3_pk(q2x - f2v{X.z, X.w}, q2y - f2v{Y.z, Y.w}, q2z - f2v{Z.z, Z.w});
          const float dd[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (dd[j] < r2) {
              buf[cnt][lane] = (unsigned int)leafno * L + g * 4 + j;
              ++cnt;
            }
          }
        }
        return tau;
      },
      [&] {  // once per round, wave converged: make room for the next bucket
        if (__any(cnt > RN_C - L)) acc.drain(buf, lane, cnt, rows, qx, qy, qz);
      });
  acc.drain(buf, lane, cnt, rows, qx, qy, qz);
  acc.finish(lq, vpx, vpy, vpz, reinterpret_cast<float4 *>(out));
}

template <int MODE>
void launch_radius_traversal(const pclmi_tree *t, bool per_lane, const float *d_q, int qsf, const int *order, int nq,
                             float r2, float tau_r, int by_index, unsigned int *counts, const u64 *offsets, u64 *keys,
                             unsigned int *n_over, hipStream_t s) {
  const unsigned waves = (unsigned)((nq + 63) / 64);
  if (per_lane) {
    const int levels = lane_nearest_levels(t->M);
    const int qpw = lane_queries_per_wave((size_t)nq, 2048);
    const unsigned lane_waves = (unsigned)((nq + qpw - 1) / qpw);
    hipLaunchKernelGGL((radius_lane_kernel<MODE>), dim3((lane_waves + 7) & ~7u), dim3(64),
                       (size_t)levels * 64 * sizeof(float), s, t->view(), d_q, qsf, order, nq, levels, r2, tau_r, by_index,
                       counts, offsets, keys, n_over, qpw);
  } else {
    const unsigned blocks = ((waves + R_WPB - 1) / R_WPB + 7) & ~7u;
    hipLaunchKernelGGL((radius_union_kernel<MODE>), dim3(blocks), dim3(64 * R_WPB), 0, s, t->view(), d_q, qsf, order, nq,
                       r2, tau_r, by_index, counts, offsets, keys, n_over);
  }
}

// ---- per-segment sort ------------------------------------------------------------------------
// One wave per segment of up to SEG_MAX keys: load into LDS (padded with ~0), bitonic network,
// store back.  Short segments (the common case, ~32 entries) take log^2 steps over 64 slots.
// Segments longer than SEG_MAX are left for the hipcub pass (their count is returned in *n_long).
constexpr int SEG_MAX = 1024;

// one single-wave block per segment, so that __syncthreads() is a wave-local barrier
__global__ __launch_bounds__(64) void segment_sort_kernel(u64 *__restrict__ keys, const u64 *__restrict__ offsets,
                                                          int nq, unsigned int *__restrict__ n_long) {
  __shared__ u64 s[SEG_MAX];
  const int lane = threadIdx.x;
  const int seg = blockIdx.x;
  const u64 a = offsets[seg], b = offsets[seg + 1];
  const u64 len64 = b - a;
  if (len64 <= 64 || len64 > (u64)SEG_MAX) return;  // block-uniform: short ones are done, long ones go to hipcub
  const int len = (int)len64;
  int P = 64;
  while (P < len) P <<= 1;
  for (int i = lane; i < P; i += 64) s[i] = i < len ? keys[a + i] : ~0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = lane; i < P / 2; i += 64) {
        const int lo = 2 * i - (i & (stride - 1));  // i-th index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const u64 x = s[lo], y = s[hi];
        const bool sw = up ? (y < x) : (x < y);
        if (sw) {
          s[lo] = y;
          s[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  for (int i = lane; i < len; i += 64) keys[a + i] = s[i];
}

// Segments of up to 64 keys (the common case: ~32 neighbours per query): one wave per segment, one key
// per lane, bitonic network through cross-lane shuffles - no LDS, no barriers.  Counts the segments it
// leaves to the other two paths: counters[0] = longer than 64, counters[1] = longer than SEG_MAX.
constexpr int SORT64_WPB = 4;
// value of lane (lane ^ STRIDE): DPP moves for the strides inside a row of 16 lanes (no LDS crossbar traffic -
// with a million short segments the ds_bpermute path is the bottleneck), ds_bpermute for 16 and 32
template <int STRIDE>
__device__ __forceinline__ unsigned int xor_lane_u32(unsigned int v) {
  if (STRIDE == 1) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
  if (STRIDE == 2) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
  if (STRIDE == 4) {
    const int m = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);                            // row_half_mirror: i ^ 7
    return (unsigned int)__builtin_amdgcn_mov_dpp(m, 0x1B, 0xF, 0xF, true);                           // quad_perm [3,2,1,0]: ^ 3
  }
  if (STRIDE == 8) return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);      // row_ror:8
  return (unsigned int)__shfl_xor((int)v, STRIDE);
}

template <int STRIDE>
__device__ __forceinline__ u64 xor_lane_u64(u64 k) {
  const unsigned int lo = xor_lane_u32<STRIDE>((unsigned int)k), hi = xor_lane_u32<STRIDE>((unsigned int)(k >> 32));
  return ((u64)hi << 32) | lo;
}

template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(u64 &key, int lane) {
  const u64 other = xor_lane_u64<STRIDE>(key);
  const bool keep_min = ((lane & SIZE) == 0) == ((lane & STRIDE) == 0);
  key = ((key < other) == keep_min) ? key : other;
  if constexpr (STRIDE > 1) bitonic_merge<SIZE, STRIDE / 2>(key, lane);
}

// bitonic network over the first P lanes of a wave (P a compile-time power of two: fully unrolled, the direction
// of every compare-exchange folds into constant lane masks)
template <int P, int SIZE = 2>
__device__ __forceinline__ void wave_bitonic_sort(u64 &key, int lane) {
  bitonic_merge<SIZE, SIZE / 2>(key, lane);
  if constexpr (SIZE < P) wave_bitonic_sort<P, SIZE * 2>(key, lane);
}

template <int P>
__device__ __forceinline__ u64 wave_bitonic(u64 key, int lane) {
  wave_bitonic_sort<P>(key, lane);
  return key;
}

constexpr int SORT64_SEGS = 8;  // consecutive segments per wave: their loads are in flight together
__global__ __launch_bounds__(64 * SORT64_WPB) void segment_sort64_kernel(u64 *__restrict__ keys,
                                                                          const u64 *__restrict__ offsets, int nq,
                                                                          unsigned int *__restrict__ counters,
                                                                          int by_index, int *__restrict__ idx_out,
                                                                          float *__restrict__ d2_out) {
  // idx_out != null: the sorted keys of the segments handled here go straight to the idx / d2 arrays
  // (same offsets); the longer segments are split by split_long_kernel after their own sort.
  // A wave takes SORT64_SEGS consecutive segments: one coalesced load of their offsets, all key loads issued
  // before the first sort (with one segment per wave the kernel was bound by wave turnover latency).
  const int lane = threadIdx.x & 63;
  const int seg0 = (blockIdx.x * SORT64_WPB + (threadIdx.x >> 6)) * SORT64_SEGS;
  if (seg0 >= nq) return;
  const int nseg = min(SORT64_SEGS, nq - seg0);
  const u64 my_off = lane <= nseg ? offsets[seg0 + lane] : 0;
  u64 a[SORT64_SEGS], key[SORT64_SEGS];
  int len[SORT64_SEGS];
  unsigned int n_gt64 = 0, n_gtmax = 0;
#pragma unroll
  for (int j = 0; j < SORT64_SEGS; ++j) {
    const u64 lo = __shfl(my_off, j), hi = __shfl(my_off, j + 1);
    a[j] = lo;
    const u64 l64 = j < nseg ? hi - lo : 0;
    if (l64 > 64) {
      ++n_gt64;
      if (l64 > (u64)SEG_MAX) ++n_gtmax;
    }
    len[j] = l64 > 64 ? 0 : (int)l64;  // longer segments are left to the other paths
    key[j] = lane < len[j] ? keys[lo + lane] : ~0ull;
  }
  if (lane == 0) {
    if (n_gt64) atomicAdd(&counters[0], n_gt64);
    if (n_gtmax) atomicAdd(&counters[1], n_gtmax);
  }
#pragma unroll
  for (int j = 0; j < SORT64_SEGS; ++j) {
    const int ln = len[j];  // wave-uniform
    if (ln == 0) continue;
    u64 k = key[j];
    // wave-uniform dispatch on the padded length: stages beyond it are never run
    if (ln > 32) k = wave_bitonic<64>(k, lane);
    else if (ln > 16) k = wave_bitonic<32>(k, lane);
    else if (ln > 8) k = wave_bitonic<16>(k, lane);
    else if (ln > 1) k = wave_bitonic<8>(k, lane);
    if (lane < ln) {
      if (idx_out) {
        const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
        idx_out[a[j] + lane] = (int)(by_index ? hi : lo);
        d2_out[a[j] + lane] = __uint_as_float(by_index ? lo : hi);
      } else {
        keys[a[j] + lane] = k;
      }
    }
  }
}

// Provisional rows (radius_union_kernel MODE 2, no query over PROV_CAP) -> sorted CSR idx / d2.
// Two consecutive segments of at most 32 keys each (most of them, at ~31 neighbours per query) share one pass of
// the 32-key network: the first in lanes 0-31, the second in lanes 32-63 - the network's last stage leaves the
// upper half in DESCENDING order, so its lanes write mirrored.
// (Measured and dropped, round 3: rows of 4-byte bucket slots with the keys recomputed here from the slot's row -
//  half the provisional bytes, traversal 384 -> 365 us, but this kernel 212 -> 298 us: it is bound by issue slots
//  and the 31M row gathers cost more than the bytes they save.)
__global__ __launch_bounds__(64 * SORT64_WPB) void prov_sort_kernel(const u64 *__restrict__ prov,
                                                                     const unsigned int *__restrict__ counts,
                                                                     const u64 *__restrict__ offsets, int nq,
                                                                     int by_index, int *__restrict__ idx_out,
                                                                     float *__restrict__ d2_out,
                                                                     const unsigned int *__restrict__ n_over, u64 cap) {
  // (enqueued before the host knows the totals: nothing to do here when a query overflowed its row or the output
  //  arrays - sized from the previous call - are too small; the host then takes the other path)
  if (*n_over != 0u || offsets[nq] > cap) return;
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int seg0 = (blockIdx.x * SORT64_WPB + (threadIdx.x >> 6)) * SORT64_SEGS;
  if (seg0 >= nq) return;
  const int nseg = min(SORT64_SEGS, nq - seg0);
  const u64 my_off = lane < nseg ? offsets[seg0 + lane] : 0;
  const unsigned int my_cnt = lane < nseg ? counts[seg0 + lane] : 0u;
  u64 a[SORT64_SEGS], key[SORT64_SEGS];
  int len[SORT64_SEGS];
#pragma unroll
  for (int j = 0; j < SORT64_SEGS; ++j) {
    a[j] = __shfl(my_off, j);
    len[j] = (int)__shfl((int)my_cnt, j);
  }
#pragma unroll
  for (int j = 0; j < SORT64_SEGS; j += 2) {
    if (len[j] <= 32 && len[j + 1] <= 32) {  // wave-uniform: the pair shares key[j]
      const int myl = half ? len[j + 1] : len[j];
      key[j] = l32 < myl ? prov[(size_t)(seg0 + j + half) * PROV_CAP + l32] : ~0ull;
      key[j + 1] = ~0ull;
    } else {
      key[j] = lane < len[j] ? prov[(size_t)(seg0 + j) * PROV_CAP + lane] : ~0ull;
      key[j + 1] = lane < len[j + 1] ? prov[(size_t)(seg0 + j + 1) * PROV_CAP + lane] : ~0ull;
    }
  }
  auto emit = [&](u64 k, u64 base, int pos) {
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    idx_out[base + pos] = (int)(by_index ? hi : lo);
    d2_out[base + pos] = __uint_as_float(by_index ? lo : hi);
  };
#pragma unroll
  for (int j = 0; j < SORT64_SEGS; j += 2) {
    if (len[j] <= 32 && len[j + 1] <= 32) {
      const int mx = max(len[j], len[j + 1]);  // wave-uniform
      if (mx == 0) continue;
      u64 k = key[j];
      if (mx > 1) k = wave_bitonic<32>(k, lane);  // lanes 0-31 ascending, lanes 32-63 descending
      const int myl = half ? len[j + 1] : len[j];
      const int pos = (half && mx > 1) ? 31 - l32 : l32;
      if (pos < myl) emit(k, half ? a[j + 1] : a[j], pos);
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ln = len[j + u];
        if (ln == 0) continue;
        u64 k = key[j + u];
        if (ln > 32) k = wave_bitonic<64>(k, lane);
        else if (ln > 16) k = wave_bitonic<32>(k, lane);
        else if (ln > 8) k = wave_bitonic<16>(k, lane);
        else if (ln > 1) k = wave_bitonic<8>(k, lane);
        if (lane < ln) emit(k, a[j + u], lane);
      }
    }
  }
}

// keys -> idx / d2 for the segments longer than 64 entries (one wave per segment)
__global__ __launch_bounds__(256) void split_long_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ offsets,
                                                         int nq, int by_index, int *__restrict__ idx,
                                                         float *__restrict__ d2) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nq) return;
  const u64 a = offsets[seg], b = offsets[seg + 1];
  if (b - a <= 64) return;
  for (u64 i = a + lane; i < b; i += 64) {
    const u64 k = keys[i];
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    idx[i] = (int)(by_index ? hi : lo);
    d2[i] = __uint_as_float(by_index ? lo : hi);
  }
}

// offsets for hipcub: segments handled by segment_sort_kernel are presented as empty
__global__ void long_segment_bounds_kernel(const u64 *__restrict__ offsets, int nq, u64 *__restrict__ beg,
                                           u64 *__restrict__ end) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const u64 a = offsets[i], b = offsets[i + 1];
  beg[i] = a;
  end[i] = (b - a > (u64)SEG_MAX) ? b : a;
}

__global__ void clamp_counts_kernel(unsigned int *__restrict__ counts, int nq, unsigned int max_nn) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) counts[i] = min(counts[i], max_nn);
}

// keys -> idx / d2 (no truncation): purely elementwise
__global__ void split_kernel(const u64 *__restrict__ keys, u64 total, int by_index, int *__restrict__ idx,
                             float *__restrict__ d2) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const u64 k = keys[i];
  const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
  idx[i] = (int)(by_index ? hi : lo);
  d2[i] = __uint_as_float(by_index ? lo : hi);
}

// truncation to the nearest max_nn: one wave per query copies the head of its (d2, index)-sorted
// segment.  to_keys != null: emit index-major keys (for the second, by-index sort) instead.
__global__ __launch_bounds__(256) void truncate_kernel(const u64 *__restrict__ keys,
                                                       const u64 *__restrict__ off_full,
                                                       const u64 *__restrict__ off_out, int nq,
                                                       int *__restrict__ idx, float *__restrict__ d2,
                                                       u64 *__restrict__ to_keys) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nq) return;
  const u64 src = off_full[seg], dst = off_out[seg];
  const int len = (int)(off_out[seg + 1] - dst);
  for (int i = lane; i < len; i += 64) {
    const u64 k = keys[src + i];
    if (to_keys) {
      to_keys[dst + i] = (k << 32) | (k >> 32);
    } else {
      idx[dst + i] = (int)(unsigned int)k;
      d2[dst + i] = __uint_as_float((unsigned int)(k >> 32));
    }
  }
}

struct CountToU64 {
  __host__ __device__ u64 operator()(unsigned int c) const { return (u64)c; }
};

// exclusive scan of counts[0 .. nq] (counts[nq] == 0) into 64-bit offsets[0 .. nq]
int scan_counts(Workspace *w, const unsigned int *d_counts, size_t nq, u64 *d_offsets, hipStream_t s) {
  hipcub::TransformInputIterator<u64, CountToU64, const unsigned int *> in(d_counts, CountToU64());
  size_t need = 0;
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, d_offsets, (int)(nq + 1), s));
  int rc = w->ensure_buf(3, need + 256);
  if (rc) return rc;
  need = w->cap_buf[3];
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(w->buf[3], need, in, d_offsets, (int)(nq + 1), s));
  return PCLMI_OK;
}

// sorts every CSR segment of `keys` ascending; `alt` is a second buffer of the same size (hipcub
// path only).  On return *sorted_in receives the buffer holding the result.
// idx_out / d2_out non-null: additionally split the sorted keys into those arrays (same offsets).
int sort_segments(Workspace *w, u64 *keys, u64 *alt, const u64 *d_offsets, size_t nq, u64 total,
                  unsigned int *d_nlong, u64 **sorted_in, hipStream_t s, int by_index = 0, int *idx_out = nullptr,
                  float *d2_out = nullptr) {
  *sorted_in = keys;
  if (total == 0) return PCLMI_OK;
  PCLMI_HIP(hipMemsetAsync(d_nlong, 0, 2 * sizeof(unsigned int), s));
  hipLaunchKernelGGL(segment_sort64_kernel,
                     dim3((unsigned)((nq + SORT64_WPB * SORT64_SEGS - 1) / (SORT64_WPB * SORT64_SEGS))),
                     dim3(64 * SORT64_WPB), 0, s, keys, d_offsets, (int)nq, d_nlong, by_index, idx_out, d2_out);
  unsigned int cnts[2] = {0, 0};  // segments longer than 64 / longer than SEG_MAX
  PCLMI_HIP(hipMemcpyAsync(cnts, d_nlong, sizeof(cnts), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  if (cnts[0] > cnts[1])
    hipLaunchKernelGGL(segment_sort_kernel, dim3((unsigned)nq), dim3(64), 0, s, keys, d_offsets, (int)nq, d_nlong);
  const unsigned int n_long = cnts[1];
  auto split_long = [&]() {
    if (idx_out && cnts[0] > 0)
      hipLaunchKernelGGL(split_long_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, (const u64 *)keys,
                         d_offsets, (int)nq, by_index, idx_out, d2_out);
  };
  if (n_long == 0) {
    split_long();
    return PCLMI_OK;
  }
  // rare: some segments exceed the LDS network; sort those with hipcub, in place via a copy
  PCLMI_CHECK(total < ((u64)1 << 31), PCLMI_ERR_UNSUPPORTED,
              "radius search: a segment longer than %d entries in a batch of more than 2^31-1 results", SEG_MAX);
  int rc = w->ensure_buf(2, nq * 2 * sizeof(u64));
  if (rc) return rc;
  u64 *beg = (u64 *)w->buf[2], *end = beg + nq;
  hipLaunchKernelGGL(long_segment_bounds_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d_offsets,
                     (int)nq, beg, end);
  PCLMI_HIP(hipMemcpyAsync(alt, keys, total * sizeof(u64), hipMemcpyDeviceToDevice, s));
  size_t need = 0;
  PCLMI_HIP(hipcub::DeviceSegmentedSort::SortKeys(nullptr, need, (const u64 *)alt, keys, (int)total, (int)nq, beg,
                                                  end, s));
  rc = w->ensure_buf(3, need + 256);
  if (rc) return rc;
  need = w->cap_buf[3];
  // hipcub leaves items outside every segment untouched in the output: short segments keep the
  // already sorted keys that are in `keys`.
  PCLMI_HIP(hipcub::DeviceSegmentedSort::SortKeys(w->buf[3], need, (const u64 *)alt, keys, (int)total, (int)nq, beg,
                                                  end, s));
  split_long();
  return PCLMI_OK;
}

}  // namespace

// the fused radius search + plane fit: d_out receives nq x (nx, ny, nz, curvature); NaN rows where no neighbour exists
int radius_normals_dev(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t qsf, float r2,
                       const float vp[3], float *d_out, hipStream_t s) {
  PCLMI_CHECK(nq < ((size_t)1 << 31) - 1, PCLMI_ERR_UNSUPPORTED, "more than 2^31-2 queries in one call");
  if (nq == 0) return PCLMI_OK;
  if (t->n == 0 || !(r2 > 0.f)) {  // no neighbour anywhere: every normal is NaN (the 0xff byte pattern is a NaN)
    PCLMI_HIP(hipMemsetAsync(d_out, 0xff, nq * 4 * sizeof(float), s));
    return PCLMI_OK;
  }
  const int32_t *order = nullptr;
  RC(launch_morton_order(t, w, d_q, nq, qsf, nullptr, nullptr, s, &order));
  const float tau_r = nextafterf(r2, 0.f);
  static const int forced = [] {
    const char *e = getenv("PCLMI_KNN");
    return !e ? 0 : (e[0] == 'u' ? 1 : (e[0] == 'l' ? 2 : 0));
  }();
  const bool per_lane = forced ? forced == 2 : nq * 4 < t->n;
  const unsigned waves = (unsigned)((nq + 63) / 64);
  TimingScope ts(PCLMI_TIMER_RADIUS, s);
  if (per_lane) {
    const int levels = lane_nearest_levels(t->M);
    const int qpw = lane_queries_per_wave(nq, 2048);
    const unsigned lane_waves = (unsigned)((nq + qpw - 1) / qpw);
    hipLaunchKernelGGL(radius_normals_lane_kernel, dim3((lane_waves + 7) & ~7u), dim3(64),
                       (size_t)levels * 64 * sizeof(float), s, t->view(), d_q, (int)qsf, order, (int)nq, levels, r2, tau_r,
                       vp[0], vp[1], vp[2], reinterpret_cast<float4 *>(d_out), qpw);
  } else {
    const unsigned blocks = ((waves + RN_WPB - 1) / RN_WPB + 7) & ~7u;
    hipLaunchKernelGGL(radius_normals_union_kernel, dim3(blocks), dim3(64 * RN_WPB), 0, s, t->view(), d_q, (int)qsf,
                       order, (int)nq, r2, tau_r, vp[0], vp[1], vp[2], reinterpret_cast<float4 *>(d_out));
  }
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

int radius_search_dev(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t qsf, float r2,
                      uint32_t max_nn, int sorted, uint64_t *d_offsets, int32_t **d_idx, float **d_d2,
                      uint64_t *total_out, hipStream_t s) {
  *d_idx = nullptr;
  *d_d2 = nullptr;
  *total_out = 0;
  PCLMI_CHECK(nq < ((size_t)1 << 31) - 1, PCLMI_ERR_UNSUPPORTED, "more than 2^31-2 queries in one call");
  if (max_nn == 0 || (size_t)max_nn > t->n) max_nn = 0;  // impl/kdtree_flann.hpp:125-135
  if (nq == 0 || t->n == 0 || !(r2 > 0.f)) {  // d2 < r2 is impossible for r2 <= 0 (or NaN)
    PCLMI_HIP(hipMemsetAsync(d_offsets, 0, (nq + 1) * sizeof(uint64_t), s));
    PCLMI_HIP(hipStreamSynchronize(s));
    return PCLMI_OK;
  }
  const int32_t *order = nullptr;
  int rc = launch_morton_order(t, w, d_q, nq, qsf, nullptr, nullptr, s, &order);
  if (rc) return rc;
  // buf[0]: counts[nq + 1] (u32) | segment-length counters[2] + overflow counter (u32) | full offsets[nq + 1]
  // (u64, only with max_nn)
  const size_t counts_bytes = ((nq + 4) * sizeof(unsigned int) + 15) & ~(size_t)15;
  rc = w->ensure_buf(0, counts_bytes + (nq + 1) * sizeof(u64));
  if (rc) return rc;
  unsigned int *d_counts = (unsigned int *)w->buf[0];
  unsigned int *d_nlong = d_counts + nq + 1;
  u64 *d_off_full = max_nn ? (u64 *)((char *)w->buf[0] + counts_bytes) : (u64 *)d_offsets;
  PCLMI_HIP(hipMemsetAsync(d_counts, 0, counts_bytes, s));
  const float tau_r = nextafterf(r2, 0.f);  // largest float < r2 (r2 > 0)
  const int by_index = (!sorted && !max_nn) ? 1 : 0;
  // sparse batch: the per-lane walk (PCLMI_KNN=union / lane forces one, as for k-NN)
  static const int forced = [] {
    const char *e = getenv("PCLMI_KNN");
    return !e ? 0 : (e[0] == 'u' ? 1 : (e[0] == 'l' ? 2 : 0));
  }();
  const bool per_lane = forced ? forced == 2 : nq * 4 < t->n;
  // One traversal when it can be enough: count and keep the first PROV_CAP keys of every query in a fixed-pitch
  // row; if no query has more, the rows are sorted into the output and the fill traversal never runs.
  unsigned int *d_nover = d_nlong + 2;
  const bool try_prov = !max_nn && nq * (size_t)PROV_CAP * sizeof(u64) <= ((size_t)2 << 30);
  u64 *prov = nullptr;
  if (try_prov) {
    // buf[1] (the CSR key arrays of the two-traversal path) doubles as the provisional rows: only one of the
    // two layouts is alive at a time (buf[3] is the scan's temporary storage and runs in between)
    rc = w->ensure_buf(1, nq * (size_t)PROV_CAP * sizeof(u64));
    if (rc) return rc;
    prov = (u64 *)w->buf[1];
  }
  {
    TimingScope ts(PCLMI_TIMER_RADIUS, s);
    if (try_prov)
      launch_radius_traversal<2>(t, per_lane, d_q, (int)qsf, order, (int)nq, r2, tau_r, by_index, d_counts,
                                 (const u64 *)nullptr, prov, d_nover, s);
    else
      launch_radius_traversal<0>(t, per_lane, d_q, (int)qsf, order, (int)nq, r2, tau_r, by_index, d_counts,
                                 (const u64 *)nullptr, (u64 *)nullptr, (unsigned int *)nullptr, s);
  }
  rc = scan_counts(w, d_counts, nq, d_off_full, s);
  if (rc) return rc;
  u64 total = 0;
  unsigned int n_over = 0;
  auto launch_prov_sort = [&](int32_t *pi, float *pd, u64 cap) {
    hipLaunchKernelGGL(prov_sort_kernel,
                       dim3((unsigned)((nq + SORT64_WPB * SORT64_SEGS - 1) / (SORT64_WPB * SORT64_SEGS))),
                       dim3(64 * SORT64_WPB), 0, s, (const u64 *)prov, (const unsigned int *)d_counts,
                       (const u64 *)d_off_full, (int)nq, by_index, pi, pd, (const unsigned int *)d_nover, cap);
  };
  // The output arrays are sized from the totals, which only the device knows at this point.  A tree remembers the
  // neighbours per query of its last radius search: with that estimate (+ 25 %) the arrays are allocated and the
  // sort is enqueued BEFORE the totals are read back - one host synchronisation per call instead of two; when the
  // estimate was too small (or a query overflowed its row) the sort does nothing and the exact path below runs.
  int32_t *spi = nullptr;
  float *spd = nullptr;
  u64 spec_cap = 0;
  if (try_prov && t->radius_hint > 0.0 && (double)nq * t->radius_hint < (double)((u64)1 << 28)) {
    spec_cap = (u64)((double)nq * t->radius_hint * 1.25) + 4096;
    spi = (int32_t *)dev_cache_get(spec_cap * sizeof(int32_t));
    spd = (float *)dev_cache_get(spec_cap * sizeof(float));
    if (spi && spd) {
      launch_prov_sort(spi, spd, spec_cap);
    } else {
      dev_cache_put(spi);
      dev_cache_put(spd);
      spi = nullptr;
      spd = nullptr;
      spec_cap = 0;
    }
  }
  auto drop_spec = [&]() {
    dev_cache_put(spi);
    dev_cache_put(spd);
    spi = nullptr;
    spd = nullptr;
  };
  {
    hipError_t e1 = hipMemcpyAsync(&total, d_off_full + nq, sizeof(u64), hipMemcpyDeviceToHost, s);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(&n_over, d_nover, sizeof(n_over), hipMemcpyDeviceToHost, s);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
    if (e1 != hipSuccess) {
      drop_spec();
      return hip_fail(e1, "radius search", __FILE__, __LINE__);
    }
  }
  t->radius_hint = nq ? (double)total / (double)nq : 0.0;
  if (total == 0) {
    drop_spec();
    if (max_nn) PCLMI_HIP(hipMemsetAsync(d_offsets, 0, (nq + 1) * sizeof(uint64_t), s));
    PCLMI_HIP(hipStreamSynchronize(s));
    return PCLMI_OK;
  }
  if (spi && n_over == 0 && total <= spec_cap) {  // the enqueued sort has done it
    *d_idx = spi;
    *d_d2 = spd;
    *total_out = total;
    return PCLMI_OK;
  }
  drop_spec();
  if (try_prov && n_over == 0) {
    int32_t *pi = (int32_t *)dev_cache_get(total * sizeof(int32_t));
    float *pd = (float *)dev_cache_get(total * sizeof(float));
    if (!pi || !pd) {
      dev_cache_put(pi);
      dev_cache_put(pd);
      return PCLMI_ERR_OOM;
    }
    launch_prov_sort(pi, pd, total);
    hipError_t pe = hipGetLastError();
    if (pe == hipSuccess) pe = hipStreamSynchronize(s);
    if (pe != hipSuccess) {
      dev_cache_put(pi);
      dev_cache_put(pd);
      return hip_fail(pe, "radius search", __FILE__, __LINE__);
    }
    *d_idx = pi;
    *d_d2 = pd;
    *total_out = total;
    return PCLMI_OK;
  }
  // buf[1]: keys[total] | alt[total]
  rc = w->ensure_buf(1, 2 * total * sizeof(u64));
  if (rc) return rc;
  u64 *keys = (u64 *)w->buf[1], *alt = keys + total;
  {
    TimingScope ts(PCLMI_TIMER_RADIUS, s);
    launch_radius_traversal<1>(t, per_lane, d_q, (int)qsf, order, (int)nq, r2, tau_r, by_index, (unsigned int *)nullptr,
                               (const u64 *)d_off_full, keys, (unsigned int *)nullptr, s);
  }
  int32_t *oi = nullptr;
  float *od = nullptr;
  auto fail = [&](int code) {
    dev_cache_put(oi);
    dev_cache_put(od);
    return code;
  };
  auto alloc_out = [&](u64 count) -> int {
    oi = (int32_t *)dev_cache_get(count * sizeof(int32_t));
    od = (float *)dev_cache_get(count * sizeof(float));
    if (!oi || !od) return fail(PCLMI_ERR_OOM);
    return PCLMI_OK;
  };
  u64 *res = nullptr;
  u64 out_total = total;
  hipError_t e;
  if (!max_nn) {
    // outputs share the CSR offsets of the keys: the sort kernels write idx / d2 directly
    if ((rc = alloc_out(total))) return rc;
    rc = sort_segments(w, keys, alt, d_off_full, nq, total, d_nlong, &res, s, by_index, oi, od);
    if (rc) return fail(rc);
  } else {
    rc = sort_segments(w, keys, alt, d_off_full, nq, total, d_nlong, &res, s);
    if (rc) return rc;
    hipLaunchKernelGGL(clamp_counts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, d_counts, (int)nq,
                       (unsigned int)max_nn);
    rc = scan_counts(w, d_counts, nq, (u64 *)d_offsets, s);
    if (rc) return rc;
    PCLMI_HIP(hipMemcpyAsync(&out_total, d_offsets + nq, sizeof(u64), hipMemcpyDeviceToHost, s));
    PCLMI_HIP(hipStreamSynchronize(s));
    if ((rc = alloc_out(out_total))) return rc;
    const unsigned seg_blocks = (unsigned)((nq + 3) / 4);
    if (sorted) {
      hipLaunchKernelGGL(truncate_kernel, dim3(seg_blocks), dim3(256), 0, s, (const u64 *)res,
                         (const u64 *)d_off_full, (const u64 *)d_offsets, (int)nq, oi, od, (u64 *)nullptr);
    } else {
      // nearest max_nn, then ascending index: truncate into index-major keys, sort again (splitting)
      u64 *other = (res == keys) ? alt : keys;
      hipLaunchKernelGGL(truncate_kernel, dim3(seg_blocks), dim3(256), 0, s, (const u64 *)res,
                         (const u64 *)d_off_full, (const u64 *)d_offsets, (int)nq, (int *)nullptr, (float *)nullptr,
                         other);
      // the source buffer's content is dead now: it serves as the spare buffer of the hipcub path
      u64 *res2 = nullptr;
      rc = sort_segments(w, other, res, (const u64 *)d_offsets, nq, out_total, d_nlong, &res2, s, 1, oi, od);
      if (rc) return fail(rc);
    }
  }
  if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess)
    return fail(hip_fail(e, "radius search", __FILE__, __LINE__));
  *d_idx = oi;
  *d_d2 = od;
  *total_out = out_total;
  return PCLMI_OK;
}

// ---- enqueue-only form (pclmi_radius_dev_async) ---------------------------------------------------------------
// What the synchronous call waits for is the output SIZE.  A caller that brings result arrays of its own - a capacity -
// does not need to know it before the kernels run: traversal into provisional rows, scan of the counts into the
// caller's offsets, sort of the rows into the caller's arrays (which does nothing when a query overflowed its row or
// the total exceeds the capacity), and a status pair the caller reads whenever it likes.  Nothing waits for the
// host, so consecutive calls queue up behind each other - or overlap on two streams, the end of one traversal under
// the start of the next.
__global__ void radius_status_kernel(const u64 *__restrict__ offsets, int nq, const unsigned int *__restrict__ n_over,
                                     u64 cap, u64 *__restrict__ status) {
  status[0] = offsets[nq];
  status[1] = (*n_over != 0u || offsets[nq] > cap) ? 1ull : 0ull;
}

int radius_search_enqueue(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t qsf, float r2,
                          int sorted, uint64_t *d_offsets, int32_t *d_idx, float *d_d2, uint64_t cap,
                          uint64_t *d_status, hipStream_t s) {
  PCLMI_CHECK(nq < ((size_t)1 << 31) - 1, PCLMI_ERR_UNSUPPORTED, "more than 2^31-2 queries in one call");
  PCLMI_CHECK(nq * (size_t)PROV_CAP * sizeof(u64) <= ((size_t)2 << 30), PCLMI_ERR_UNSUPPORTED,
              "more than %zu queries in one enqueue-only radius search", ((size_t)2 << 30) / (PROV_CAP * sizeof(u64)));
  if (nq == 0 || t->n == 0 || !(r2 > 0.f)) {
    PCLMI_HIP(hipMemsetAsync(d_offsets, 0, (nq + 1) * sizeof(uint64_t), s));
    PCLMI_HIP(hipMemsetAsync(d_status, 0, 2 * sizeof(uint64_t), s));
    return PCLMI_OK;
  }
  const int32_t *order = nullptr;
  RC(launch_morton_order(t, w, d_q, nq, qsf, nullptr, nullptr, s, &order));
  const size_t counts_bytes = ((nq + 4) * sizeof(unsigned int) + 15) & ~(size_t)15;
  RC(w->ensure_buf(0, counts_bytes));
  RC(w->ensure_buf(1, nq * (size_t)PROV_CAP * sizeof(u64)));
  unsigned int *d_counts = (unsigned int *)w->buf[0];
  unsigned int *d_nover = d_counts + nq + 3;
  u64 *prov = (u64 *)w->buf[1];
  PCLMI_HIP(hipMemsetAsync(d_counts, 0, counts_bytes, s));
  const float tau_r = nextafterf(r2, 0.f);
  const int by_index = sorted ? 0 : 1;
  static const int forced = [] {
    const char *e = getenv("PCLMI_KNN");
    return !e ? 0 : (e[0] == 'u' ? 1 : (e[0] == 'l' ? 2 : 0));
  }();
  const bool per_lane = forced ? forced == 2 : nq * 4 < t->n;
  {
    TimingScope ts(PCLMI_TIMER_RADIUS, s);
    launch_radius_traversal<2>(t, per_lane, d_q, (int)qsf, order, (int)nq, r2, tau_r, by_index, d_counts,
                               (const u64 *)nullptr, prov, d_nover, s);
  }
  RC(scan_counts(w, d_counts, nq, (u64 *)d_offsets, s));
  hipLaunchKernelGGL(prov_sort_kernel, dim3((unsigned)((nq + SORT64_WPB * SORT64_SEGS - 1) / (SORT64_WPB * SORT64_SEGS))),
                     dim3(64 * SORT64_WPB), 0, s, (const u64 *)prov, (const unsigned int *)d_counts,
                     (const u64 *)d_offsets, (int)nq, by_index, d_idx, d_d2, (const unsigned int *)d_nover, (u64)cap);
  hipLaunchKernelGGL(radius_status_kernel, dim3(1), dim3(1), 0, s, (const u64 *)d_offsets, (int)nq,
                     (const unsigned int *)d_nover, (u64)cap, (u64 *)d_status);
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

// ---- a few queries from the host (pcl::KdTreeFLANN::radiusSearch, one point per call) --------------------------
// The general path above is built for batches: ordering pass, count scan, totals read back, output arrays from the
// cache, three downloads - a dozen enqueues and two host waits, 133 us for one query (204 for 1024).  Up to kSmallRadiusQueries
// queries take two launches and ONE wait instead: the per-lane traversal leaves every query's first PROV_CAP keys in a
// fixed-pitch row (device memory) and its count in the slot's pinned host buffer; one wavefront per query then sorts
// the row and writes idx / d2 rows into the pinned buffer in place.  The host builds the CSR arrays from there.  A
// query with more than PROV_CAP neighbours sends the call to the general path.
constexpr size_t kSmallRadiusQueries = 1024;
__global__ __launch_bounds__(64) void small_rows_sort_kernel(const u64 *__restrict__ prov,
                                                             const unsigned int *__restrict__ counts, int nq,
                                                             int by_index, int *__restrict__ idx_rows,
                                                             float *__restrict__ d2_rows) {
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (qi >= nq) return;
  const int ln = (int)counts[qi];
  if (ln == 0 || ln > PROV_CAP) return;  // (too long: the host takes the general path)
  u64 k = lane < ln ? prov[(size_t)qi * PROV_CAP + lane] : ~0ull;
  if (ln > 32) k = wave_bitonic<64>(k, lane);
  else if (ln > 16) k = wave_bitonic<32>(k, lane);
  else if (ln > 8) k = wave_bitonic<16>(k, lane);
  else if (ln > 1) k = wave_bitonic<8>(k, lane);
  if (lane < ln) {
    const unsigned int hi = (unsigned int)(k >> 32), lo = (unsigned int)k;
    idx_rows[(size_t)qi * PROV_CAP + lane] = (int)(by_index ? hi : lo);
    d2_rows[(size_t)qi * PROV_CAP + lane] = __uint_as_float(by_index ? lo : hi);
  }
}

// returns PCLMI_OK with *done = true when the call was served here
static int radius_small_host(const pclmi_tree *t, HostSlot *h, const float *q, size_t nq, size_t q_stride_bytes,
                             float r2, uint32_t max_nn, int sorted, uint64_t *offsets, int32_t **idx, float **d2,
                             bool *done) {
  *done = false;
  static const bool off = [] { const char *e = getenv("PCLMI_RADIUS_SMALL"); return e && e[0] == '0'; }();
  if (off || nq == 0 || nq > kSmallRadiusQueries || t->n == 0 || !(r2 > 0.f)) return PCLMI_OK;
  if (max_nn == 0 || (size_t)max_nn > t->n) max_nn = 0;  // impl/kdtree_flann.hpp:125-135
  const size_t q_bytes = nq * q_stride_bytes;
  const size_t off_c = (q_bytes + 255) & ~(size_t)255;
  const size_t off_i = off_c + ((nq * sizeof(unsigned int) + 255) & ~(size_t)255);
  const size_t off_d = off_i + nq * PROV_CAP * sizeof(int32_t);
  if (off_d + nq * PROV_CAP * sizeof(float) > HostSlot::kPinBytes) return PCLMI_OK;
  RC(h->ensure(1, nq * (size_t)PROV_CAP * sizeof(u64) + 256));
  u64 *prov = (u64 *)h->d[1];
  unsigned int *d_scrap = (unsigned int *)(prov + nq * PROV_CAP);  // the traversal's overflow counter: not used here
  std::memcpy(h->h_pin, q, q_bytes);
  unsigned int *counts_h = (unsigned int *)(h->h_pin + off_c);
  std::memset(counts_h, 0, nq * sizeof(unsigned int));  // a query with non-finite coordinates writes 0 anyway
  const float tau_r = nextafterf(r2, 0.f);
  // nearest-first keys unless the caller wants the plain unsorted list (ascending index, as the general path gives it)
  const int by_index = (!sorted && !max_nn) ? 1 : 0;
  launch_radius_traversal<2>(t, true, (const float *)h->d_pin, (int)(q_stride_bytes / 4), nullptr, (int)nq, r2, tau_r,
                             by_index, (unsigned int *)(h->d_pin + off_c), (const u64 *)nullptr, prov, d_scrap, h->s);
  hipLaunchKernelGGL(small_rows_sort_kernel, dim3((unsigned)nq), dim3(64), 0, h->s, (const u64 *)prov,
                     (const unsigned int *)(h->d_pin + off_c), (int)nq, by_index, (int *)(h->d_pin + off_i),
                     (float *)(h->d_pin + off_d));
  PCLMI_HIP(hipGetLastError());
  PCLMI_HIP(hipStreamSynchronize(h->s));
  uint64_t total = 0;
  for (size_t i = 0; i < nq; ++i) {
    if (counts_h[i] > (unsigned int)PROV_CAP) return PCLMI_OK;  // the general path
    offsets[i] = total;
    total += max_nn ? std::min<uint64_t>(counts_h[i], max_nn) : counts_h[i];
  }
  offsets[nq] = total;
  int32_t *hi = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));
  float *hd = (float *)malloc((total ? total : 1) * sizeof(float));
  if (!hi || !hd) {
    free(hi);
    free(hd);
    set_error("out of host memory for %llu radius results", (unsigned long long)total);
    return PCLMI_ERR_OOM;
  }
  const int32_t *ri = (const int32_t *)(h->h_pin + off_i);
  const float *rd = (const float *)(h->h_pin + off_d);
  for (size_t i = 0; i < nq; ++i) {
    const size_t m = (size_t)(offsets[i + 1] - offsets[i]);
    int32_t *oi = hi + offsets[i];
    float *od = hd + offsets[i];
    std::memcpy(oi, ri + i * PROV_CAP, m * sizeof(int32_t));
    std::memcpy(od, rd + i * PROV_CAP, m * sizeof(float));
    if (max_nn && !sorted && m > 1) {  // the nearest max_nn, listed by ascending index (as the general path does)
      std::pair<int32_t, float> tmp[PROV_CAP];
      for (size_t j = 0; j < m; ++j) tmp[j] = {oi[j], od[j]};
      std::sort(tmp, tmp + m, [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) {
        return a.first < b.first;
      });
      for (size_t j = 0; j < m; ++j) {
        oi[j] = tmp[j].first;
        od[j] = tmp[j].second;
      }
    }
  }
  *idx = hi;
  *d2 = hd;
  *done = true;
  return PCLMI_OK;
}

}  // namespace pclmi

using namespace pclmi;

extern "C" {

int pclmi_radius_dev(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, double radius,
                     uint32_t max_nn, int sorted, uint64_t *d_offsets, int32_t **d_idx, float **d_d2,
                     uint64_t *total, void *stream) {
  PCLMI_CHECK(t && d_offsets && d_idx && d_d2 && total, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(nq == 0 || d_q, PCLMI_ERR_INVALID, "queries are NULL");
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  const float r2 = (float)(radius * radius);  // impl/kdtree_flann.hpp:140
  WsGuard g(t, s);
  return radius_search_dev(t, g.w, d_q, nq, q_stride_bytes / 4, r2, max_nn, sorted, d_offsets, d_idx, d_d2, total,
                           s);
}

int pclmi_radius_dev_async(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, double radius,
                           int sorted, uint64_t *d_offsets, int32_t *d_idx, float *d_d2, uint64_t capacity,
                           uint64_t *d_status, void *stream) {
  PCLMI_CHECK(t && d_offsets && d_status, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(nq == 0 || d_q, PCLMI_ERR_INVALID, "queries are NULL");
  PCLMI_CHECK(capacity == 0 || (d_idx && d_d2), PCLMI_ERR_INVALID, "NULL result arrays");
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  Workspace *w = acquire_ws(t, s);
  const int rc = radius_search_enqueue(t, w, d_q, nq, q_stride_bytes / 4, (float)(radius * radius), sorted, d_offsets,
                                       d_idx, d_d2, capacity, d_status, s);
  release_ws_async(t, w, s);  // the next user of w is ordered behind this call's kernels by an event
  return rc;
}

int pclmi_radius(const pclmi_tree *t, const float *q, size_t nq, size_t q_stride_bytes, double radius,
                 uint32_t max_nn, int sorted, uint64_t *offsets, int32_t **idx, float **d2) {
  PCLMI_CHECK(t && offsets && idx && d2, PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(nq == 0 || q, PCLMI_ERR_INVALID, "queries are NULL");
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  *idx = nullptr;
  *d2 = nullptr;
  PCLMI_HIP(hipSetDevice(t->device));
  // queries and offsets are staged in the tree's pooled host-path buffers (no hipMalloc / hipFree in the steady
  // state); a few queries are read by the kernels straight from the slot's pinned host buffer
  HostSlotGuard hg(t);
  if (!hg.h) return PCLMI_ERR_HIP;
  HostSlot *h = hg.h;
  {
    bool done = false;
    RC(radius_small_host(t, h, q, nq, q_stride_bytes, (float)(radius * radius), max_nn, sorted, offsets, idx, d2, &done));
    if (done) return PCLMI_OK;
  }
  int32_t *d_idx = nullptr;
  float *d_d2 = nullptr;
  uint64_t total = 0;
  const size_t q_bytes = nq * q_stride_bytes;
  const float *d_q = nullptr;
  if (q_bytes <= HostSlot::kPinBytes) {
    if (nq) std::memcpy(h->h_pin, q, q_bytes);
    d_q = (const float *)h->d_pin;
  } else {
    RC(h->ensure(0, q_bytes));
    PCLMI_HIP(hipMemcpyAsync(h->d[0], q, q_bytes, hipMemcpyHostToDevice, h->s));
    d_q = (const float *)h->d[0];
  }
  RC(h->ensure(3, (nq + 1) * sizeof(uint64_t)));
  uint64_t *d_off = (uint64_t *)h->d[3];
  int rc = pclmi_radius_dev(t, d_q, nq, q_stride_bytes, radius, max_nn, sorted, d_off, &d_idx, &d_d2, &total, h->s);
  hipError_t e;
  if (rc == PCLMI_OK) {
    int32_t *hi = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));
    float *hd = (float *)malloc((total ? total : 1) * sizeof(float));
    if (!hi || !hd) {
      free(hi);
      free(hd);
      set_error("out of host memory for %llu radius results", (unsigned long long)total);
      rc = PCLMI_ERR_OOM;
    } else if ((e = hipMemcpyAsync(offsets, d_off, (nq + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, h->s)) != hipSuccess ||
               (total && ((e = hipMemcpyAsync(hi, d_idx, total * sizeof(int32_t), hipMemcpyDeviceToHost, h->s)) != hipSuccess ||
                          (e = hipMemcpyAsync(hd, d_d2, total * sizeof(float), hipMemcpyDeviceToHost, h->s)) != hipSuccess)) ||
               (e = hipStreamSynchronize(h->s)) != hipSuccess) {
      free(hi);
      free(hd);
      rc = hip_fail(e, "result download", __FILE__, __LINE__);
    } else {
      *idx = hi;
      *d2 = hd;
    }
  }
  pclmi_free_dev(d_idx);  // result buffers belong to the library's cache (never hipFree them directly)
  pclmi_free_dev(d_d2);
  return rc;
}

}  // extern "C"
