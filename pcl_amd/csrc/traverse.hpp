// traverse.hpp - device-side building blocks of the wave-uniform ("union") tree traversal,
// shared by the kNN, radius and ICP kernels.  gfx950 / wave64 only.
#ifndef PCLMI_TRAVERSE_HPP_
#define PCLMI_TRAVERSE_HPP_

#include "pclmi_internal.h"

namespace pclmi {
namespace dev {

// Uniform (scalar) loads: reading through the constant address space with a wave-uniform
// address makes hipcc emit s_load_dwordx8 / x16 into SGPRs instead of per-lane vector loads.
typedef float f8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const f8v __attribute__((address_space(4))) *cbox_ptr;    // one node box (32 B)
typedef const f16v __attribute__((address_space(4))) *cgrp_ptr;   // 64-byte line

// the pre-pass (lane_bound) scans PCLMI_PREPASS_MUL * K points around the query's own bucket (a power of two;
// measured at 1M x 1M, k = 16: 2 -> 0.765 ms, 4 -> 0.735 ms, 8 -> 0.755 ms)
#ifndef PCLMI_PREPASS_MUL
#define PCLMI_PREPASS_MUL 4
#endif
#ifndef PCLMI_ASM_MIN
#define PCLMI_ASM_MIN 1
#endif
// pre-pass list building by sorting network + bitonic merge (K >= 16) instead of the insertion chain
#ifndef PCLMI_PREPASS_SORT
#define PCLMI_PREPASS_SORT 1
#endif
#ifndef PCLMI_PREPASS_DESCENT2
#define PCLMI_PREPASS_DESCENT2 1
#endif
#ifndef PCLMI_ASM_CHAIN
#define PCLMI_ASM_CHAIN 1
#endif
constexpr int L = kLeafSize;
constexpr int G = L / 4;

// point groups per bucket: [x0..3 | y0..3 | z0..3 | idx0..3]

__device__ __forceinline__ float sq3(float dx, float dy, float dz) {
  // one rounding per operation, fixed order: ((dx*dx + dy*dy) + dz*dz)
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// squared distance from q to an axis-aligned box given as 8 uniform floats.  Uses the same
// operation order as sq3, so it is a true lower bound (in fp32) of sq3 to any point inside.
__device__ __forceinline__ float box_dist(float qx, float qy, float qz, float lx, float ly, float lz,
                                          float hx, float hy, float hz) {
  float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
  float dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
  float dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
  return sq3(dx, dy, dz);
}

// Packed fp32 (v_pk_add_f32 / v_pk_mul_f32): two candidates or two boxes per instruction.  Each
// half is an ordinary IEEE fp32 operation, so results are bit-identical to sq3 / box_dist.
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2v sq3_pk(f2v dx, f2v dy, f2v dz) {
#pragma clang fp contract(off)
  const f2v xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const f2v s = xx + yy;
  return s + zz;
}

__device__ __forceinline__ f2v box_dist_pk(f2v qx, f2v qy, f2v qz, f2v lx, f2v ly, f2v lz, f2v hx, f2v hy,
                                           f2v hz) {
#pragma clang fp contract(off)
  const f2v ax = lx - qx, bx = qx - hx;
  const f2v ay = ly - qy, by = qy - hy;
  const f2v az = lz - qz, bz = qz - hz;
  f2v dx, dy, dz;
  dx.x = fmaxf(fmaxf(ax.x, bx.x), 0.f);
  dx.y = fmaxf(fmaxf(ax.y, bx.y), 0.f);
  dy.x = fmaxf(fmaxf(ay.x, by.x), 0.f);
  dy.y = fmaxf(fmaxf(ay.y, by.y), 0.f);
  dz.x = fmaxf(fmaxf(az.x, bz.x), 0.f);
  dz.y = fmaxf(fmaxf(az.y, bz.y), 0.f);
  return sq3_pk(dx, dy, dz);
}

__device__ __forceinline__ unsigned long long make_key(float d, int i) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)i;
}

template <int K>
__device__ __forceinline__ void list_insert(float (&a)[K], float d) {
  // a ascending.  new a[j] = median(a[j-1], a[j], d); new a[0] = min(a[0], d).
#if PCLMI_ASM_CHAIN
  // the same instructions with the destination TIED to a[j]: the list is updated in place, in descending j.  Left to
  // itself the compiler gives the updated list registers of its own and copies all K values across at every bucket
  // of the k-NN walk (twice: in and out), some 25 moves per bucket.
#pragma unroll
  for (int j = K - 1; j >= 1; --j) asm("v_med3_f32 %0, %1, %0, %2" : "+v"(a[j]) : "v"(a[j - 1]), "v"(d));
#else
#pragma unroll
  for (int j = K - 1; j >= 1; --j) a[j] = __builtin_amdgcn_fmed3f(a[j - 1], a[j], d);
#endif
#if PCLMI_ASM_MIN
  // the compiler folds med3(-inf, a, d) into a minnum with a canonicalising v_max on either operand (3 instructions);
  // the bare instruction returns the same value for every input the list sees (a NaN d leaves a[0] as it is)
  asm("v_min_f32 %0, %0, %1" : "+v"(a[0]) : "v"(d));
#else
  a[0] = __builtin_amdgcn_fmed3f(-__builtin_inff(), a[0], d);
#endif
}

// Per-lane list building for the pre-pass, where EVERY lane has 16 new distances per bucket: sort the bucket's
// distances with a 63-exchange network (Batcher's odd-even merge sort, OddEvenMergeNet), keep the K smallest of list and bucket with a
// half-cleaner (a[K-16+i] = min(a[K-16+i], b[15-i]): a bitonic sequence) and sort that with the log2(K) stages of a
// bitonic merger.  K = 16: 286 instructions per bucket instead of the chain's 272 + ..., see lane_bound_list.
__device__ __forceinline__ void exchange_asc(float &x, float &y) {  // x <- min, y <- max (bare instructions: no NaNs here)
  float lo, hi;
  asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
  x = lo;
  y = hi;
}

// Batcher's odd-even merge sort for N = 2^m inputs as a list of compare-exchange pairs (i < j: the smaller value goes
// to i), generated at compile time: 63 exchanges for 16 inputs, 191 for 32, 543 for 64 (a bitonic sorter: 80 / 240 / 672).
template <int N>
struct OddEvenMergeNet {
  static_assert(N >= 2 && N <= 64 && (N & (N - 1)) == 0, "power of two");
  unsigned char lo[N * 9], hi[N * 9];
  int n;
  constexpr OddEvenMergeNet() : lo{}, hi{}, n(0) {
    for (int p = 1; p < N; p <<= 1)
      for (int k = p; k >= 1; k >>= 1)
        for (int j = k % p; j + k < N; j += 2 * k)
          for (int i = 0; i < k && i + j + k < N; ++i)
            if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
              lo[n] = (unsigned char)(i + j);
              hi[n] = (unsigned char)(i + j + k);
              ++n;
            }
  }
};

__device__ __forceinline__ void sort16_asc(float (&b)[16]) {
  constexpr OddEvenMergeNet<16> net{};
#pragma unroll
  for (int c = 0; c < net.n; ++c) exchange_asc(b[net.lo[c]], b[net.hi[c]]);
}

// a (ascending, K >= 16 a power of two) <- the K smallest of a and b (ascending), ascending
template <int K>
__device__ __forceinline__ void list_merge16(float (&a)[K], const float (&b)[16]) {
  static_assert(K >= 16 && (K & (K - 1)) == 0, "power of two");
#pragma unroll
  for (int i = 0; i < 16; ++i) asm("v_min_f32 %0, %0, %1" : "+v"(a[K - 16 + i]) : "v"(b[15 - i]));
#pragma unroll
  for (int d = K / 2; d >= 1; d >>= 1) {
#pragma unroll
    for (int i = 0; i < K; ++i)
      if ((i & d) == 0) exchange_asc(a[i], a[i + d]);
  }
}

// wave-uniform stack kept in the lanes of one VGPR
struct LaneStack {
  int v;
  int sp;
  int lane;
  __device__ __forceinline__ void push(int node) {
    v = (lane == sp) ? node : v;
    ++sp;
  }
  __device__ __forceinline__ int pop() {
    --sp;
    return __builtin_amdgcn_readlane(v, sp);
  }
};

// ---------------------------------------------------------------------------------------------
// Shared prologue: XCD-aware wave -> query-chunk assignment and query load.
struct LaneQuery {
  float x, y, z;
  int qi;
  int slot;         // position in the sorted batch
  bool alive;       // lane maps to a query row
  bool searchable;  // ... and that query is finite
};

// qpw: queries per wavefront (lanes qpw..63 stay idle).  Small batches of the per-lane kernels are spread over more
// wavefronts than they would fill: a wavefront's time is the LONGEST of its lanes' walks, and an empty GPU has slots.
template <int WPB>
__device__ __forceinline__ bool load_lane_query(const float *__restrict__ q, int q_stride_f,
                                                const int *__restrict__ order, int nq, LaneQuery &o, int qpw = 64) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  // blocks b, b+8, b+16 ... share an XCD (and its L2): give each XCD one contiguous range of the
  // sorted queries so that its L2 holds one part of the tree.  (Placement only affects speed.)
  const int per = (gridDim.x + 7) >> 3;
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int slot = (blk * WPB + wib) * qpw + lane;
  if (__builtin_amdgcn_readfirstlane(slot) >= nq) return false;  // whole wave past the end
  o.alive = slot < nq && lane < qpw;
  o.slot = slot;
  o.qi = o.alive ? (order ? order[slot] : slot) : 0;
  o.x = o.y = o.z = __builtin_nanf("");  // dead lanes: every comparison is false
  if (o.alive) {
    const float *p = q + (size_t)o.qi * q_stride_f;
    o.x = p[0];
    o.y = p[1];
    o.z = p[2];
  }
  // Lanes past the end and non-finite queries must never steer the wave: their coordinates are
  // NaN (no candidate ever passes) and the callers give them a NEGATIVE bound, because
  // box_dist(NaN) is 0 (fmax drops NaNs) and would otherwise "need" every box of the tree.
  o.searchable = o.alive && isfinite(o.x) && isfinite(o.y) && isfinite(o.z);
  if (!o.searchable) o.x = o.y = o.z = __builtin_nanf("");
  return true;
}

// Queries per wavefront of a per-lane kernel: 64 once the batch fills `resident` wavefronts (what the GPU holds at a
// time), else the power of two that spreads it over about that many (see load_lane_query).  PCLMI_LANE_WAVES overrides
// `resident` (development knob; 0 = always 64).
inline int lane_queries_per_wave(size_t nq, int resident) {
  static const int env = [] {
    const char *e = getenv("PCLMI_LANE_WAVES");
    return e ? atoi(e) : -1;
  }();
  if (env >= 0) resident = env;
  int qpw = 64;
  while (qpw > 1 && (nq + qpw / 2 - 1) / (qpw / 2) <= (size_t)resident) qpw >>= 1;
  return qpw;
}

// Per-lane pre-pass: walk down to a nearby bucket (towards the nearer child box), scan it and its
// sibling bucket with ordinary vector loads, and return the K-th smallest distance found.  This
// is only an UPPER BOUND on the true K-th distance (FLT_MAX when fewer than K points were seen);
// starting the union traversal with it prunes the far subtrees from the first step on.
// lane_bound_list leaves the whole ascending list of the K smallest distances found in a[] and the first of the NB
// sibling buckets scanned (prepass_buckets<K>() of them) in home - the union k-NN kernel keeps both, see there.
template <int K>
__host__ __device__ constexpr int prepass_buckets() {
  return (PCLMI_PREPASS_MUL * K / L) > 2 ? (PCLMI_PREPASS_MUL * K / L) : 2;
}

template <int K>
__device__ __forceinline__ void lane_bound_list(const TreeView &t, float qx, float qy, float qz, float (&a)[K],
                                                int &home) {
  const int M = t.M;
  int node = 1;
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#if PCLMI_PREPASS_DESCENT2
  // two levels per dependent load (the four grandchildren boxes are two adjacent 64-byte lines), packed box distances:
  // towards the nearest grandchild.  Any bucket near the query will do - the walk only wants a finite starting bound.
  while (2 * node + 1 < M) {
    const float4 *b = reinterpret_cast<const float4 *>(t.box + (size_t)node * 32);
    const float4 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3], c0 = b[4], c1 = b[5], c2 = b[6], c3 = b[7];
    const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{a0.x, a2.x}, f2v{a0.y, a2.y}, f2v{a0.z, a2.z}, f2v{a0.w, a2.w},
                                f2v{a1.x, a3.x}, f2v{a1.y, a3.y});
    const f2v m23 = box_dist_pk(q2x, q2y, q2z, f2v{c0.x, c2.x}, f2v{c0.y, c2.y}, f2v{c0.z, c2.z}, f2v{c0.w, c2.w},
                                f2v{c1.x, c3.x}, f2v{c1.y, c3.y});
    const bool r01 = m01.y < m01.x, r23 = m23.y < m23.x;
    const float lo01 = r01 ? m01.y : m01.x, lo23 = r23 ? m23.y : m23.x;
    const bool right = lo23 < lo01;
    node = 4 * node + (right ? 2 + (r23 ? 1 : 0) : (r01 ? 1 : 0));
  }
#endif
  while (node < M) {
    const float4 *b = reinterpret_cast<const float4 *>(t.box + (size_t)node * 16);
    const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    const float m0 = box_dist(qx, qy, qz, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y);
    const float m1 = box_dist(qx, qy, qz, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y);
    node = 2 * node + (m1 < m0 ? 1 : 0);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) a[j] = 3.402823466e+38f;
  // scan the NB sibling buckets around the one reached (>= 2K points when they are all full)
  constexpr int NB = prepass_buckets<K>();
  const int first = node & ~(NB - 1);
  home = first;
#pragma unroll 1
  for (int s = 0; s < NB; ++s) {
    const int bn = first + s;
    if (bn < M || bn > 2 * M - 1) continue;
    const float4 *rec = reinterpret_cast<const float4 *>(t.leaf + (size_t)(bn - M) * (4 * L));
    float b[L];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 X = rec[g * 4 + 0], Y = rec[g * 4 + 1], Z = rec[g * 4 + 2];
      const f2v d01 = sq3_pk(q2x - f2v{X.x, X.y}, q2y - f2v{Y.x, Y.y}, q2z - f2v{Z.x, Z.y});
      const f2v d23 = sq3_pk(q2x - f2v{X.z, X.w}, q2y - f2v{Y.z, Y.w}, q2z - f2v{Z.z, Z.w});
      b[4 * g + 0] = d01.x;
      b[4 * g + 1] = d01.y;
      b[4 * g + 2] = d23.x;
      b[4 * g + 3] = d23.y;
    }
    if constexpr (PCLMI_PREPASS_SORT && K >= 16 && (K & (K - 1)) == 0 && L == 16) {
      sort16_asc(b);
      list_merge16<K>(a, b);
    } else {
#pragma unroll
      for (int j = 0; j < L; ++j) list_insert<K>(a, b[j]);
    }
  }
}

template <int K>
__device__ __forceinline__ float lane_bound(const TreeView &t, float qx, float qy, float qz) {
  float a[K];
  int home;
  lane_bound_list<K>(t, qx, qy, qz, a, home);
  return a[K - 1];
}

// One step of the wave-uniform traversal at inner node `node`: tests the 4 grandchildren at once
// when both children are inner nodes (their boxes are two adjacent 64-byte lines), else the 2
// children; pushes the boxes some lane still needs, nearest (by lane majority) on top.
__device__ __forceinline__ void expand_inner(LaneStack &st, const cgrp_ptr box2, int node, int M, float qx,
                                             float qy, float qz, float tau) {
  if (2 * node + 1 < M) {
    const f16v ba = box2[2 * node], bb = box2[2 * node + 1];
    const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
    const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{ba[0], ba[8]}, f2v{ba[1], ba[9]}, f2v{ba[2], ba[10]},
                                f2v{ba[3], ba[11]}, f2v{ba[4], ba[12]}, f2v{ba[5], ba[13]});
    const f2v m23 = box_dist_pk(q2x, q2y, q2z, f2v{bb[0], bb[8]}, f2v{bb[1], bb[9]}, f2v{bb[2], bb[10]},
                                f2v{bb[3], bb[11]}, f2v{bb[4], bb[12]}, f2v{bb[5], bb[13]});
    const float m0 = m01.x, m1 = m01.y, m2 = m23.x, m3 = m23.y;
    const bool n0 = __any(m0 <= tau), n1 = __any(m1 <= tau), n2 = __any(m2 <= tau), n3 = __any(m3 <= tau);
    const bool a_first = __popcll(__ballot(fminf(m0, m1) <= fminf(m2, m3))) >= 32;
    const bool f01 = __popcll(__ballot(m0 <= m1)) >= 32;
    const bool f23 = __popcll(__ballot(m2 <= m3)) >= 32;
    const int g = 4 * node;
#define PCLMI_PUSH_PAIR(base, lowfirst, nlo, nhi) \
  do {                                            \
    if (lowfirst) {                               \
      if (nhi) st.push((base) + 1);               \
      if (nlo) st.push(base);                     \
    } else {                                      \
      if (nlo) st.push(base);                     \
      if (nhi) st.push((base) + 1);               \
    }                                             \
  } while (0)
    if (a_first) {
      PCLMI_PUSH_PAIR(g + 2, f23, n2, n3);
      PCLMI_PUSH_PAIR(g, f01, n0, n1);
    } else {
      PCLMI_PUSH_PAIR(g, f01, n0, n1);
      PCLMI_PUSH_PAIR(g + 2, f23, n2, n3);
    }
  } else {
    const f16v b = box2[node];
    const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
    const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{b[0], b[8]}, f2v{b[1], b[9]}, f2v{b[2], b[10]},
                                f2v{b[3], b[11]}, f2v{b[4], b[12]}, f2v{b[5], b[13]});
    const float m0 = m01.x, m1 = m01.y;
    const bool n0 = __any(m0 <= tau), n1 = __any(m1 <= tau);
    const bool f01 = __popcll(__ballot(m0 <= m1)) >= 32;
    PCLMI_PUSH_PAIR(2 * node, f01, n0, n1);
  }
#undef PCLMI_PUSH_PAIR
}


// Wave-uniform 1-NN search: returns each lane's best 64-bit key (d2 bits << 32 | index) for the
// lane's query; non-searchable lanes return the initial key (FLT_MAX, 0x7fffffff).
// tau_known: the caller has an upper bound per lane (tau0, attained by an indexed point) and the node every lane's
// ball lies under (root, e.g. wave_common_root); otherwise the bound comes from the per-lane pre-pass and the walk
// starts at the tree's root.  best_slot (nullable): bucket slot of the result.
__device__ __forceinline__ unsigned long long wave_nearest(const TreeView &t, float qx, float qy, float qz,
                                                           bool searchable, int lane, bool tau_known = false,
                                                           float tau0 = 0.f, int root = 1, int *best_slot = nullptr) {
  unsigned long long best = make_key(3.402823466e+38f, 0x7fffffff);
  int bslot = 0;
  float tau = !searchable ? -1.f : tau_known ? tau0 : lane_bound<1>(t, qx, qy, qz);
  LaneStack st{0, 0, lane};
  st.push(tau_known ? root : 1);
  const cbox_ptr box = (cbox_ptr)t.box;
  const cgrp_ptr box2 = (cgrp_ptr)t.box;
  const cgrp_ptr leaf = (cgrp_ptr)t.leaf;
  const int M = t.M;
  while (st.sp > 0) {
    const int node = st.pop();
    if (node >= M) {
      const f8v b = box[node];
      const cgrp_ptr rec = leaf + (size_t)(node - M) * G;
      f16v grp[G];
#pragma unroll
      for (int g = 0; g < G; ++g) grp[g] = rec[g];
      if (!__any(box_dist(qx, qy, qz, b[0], b[1], b[2], b[3], b[4], b[5]) <= tau)) continue;
      const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f2v d2 = sq3_pk(q2x - f2v{grp[g][2 * h], grp[g][2 * h + 1]},
                                q2y - f2v{grp[g][4 + 2 * h], grp[g][5 + 2 * h]},
                                q2z - f2v{grp[g][8 + 2 * h], grp[g][9 + 2 * h]});
          const unsigned long long k0 = make_key(d2.x, __float_as_int(grp[g][12 + 2 * h]));
          const unsigned long long k1 = make_key(d2.y, __float_as_int(grp[g][13 + 2 * h]));
          const int s0 = (node - M) * L + g * 4 + 2 * h;
          bslot = k0 < best ? s0 : bslot;
          best = k0 < best ? k0 : best;
          bslot = k1 < best ? s0 + 1 : bslot;
          best = k1 < best ? k1 : best;
        }
      }
      tau = fminf(tau, __uint_as_float((unsigned int)(best >> 32)));
    } else {
      expand_inner(st, box2, node, M, qx, qy, qz, tau);
    }
  }
  if (best_slot) *best_slot = bslot;
  return best;
}

// ---------------------------------------------------------------------------------------------
// Per-lane exact 1-NN: every lane walks the tree for its OWN query - near child first, the far
// child's box distance remembered per level in LDS (far[level][lane]), stackless backtracking through
// the implicit heap order (parent = id >> 1, sibling = id ^ 1, one "second child taken" bit per level).
// With a good starting bound (ICP: the distance to the previous iteration's neighbour) a lane touches
// one root-to-leaf path and a leaf or two, ~4x fewer VALU instructions than the wave-uniform walk
// above; neighbouring lanes follow neighbouring paths, so the vector loads hit the same L1/L2 lines.
// tau0: any upper bound of the lane's nearest distance that is attained by an indexed point (+inf when
// none is known).  far: this wave's LDS scratch with lane_nearest_levels(M) rows.  Returns the best
// 64-bit key (d2 bits << 32 | index); lanes with searchable == false return (FLT_MAX, 0x7fffffff).
__host__ __device__ inline int lane_nearest_levels(int M) {
  int lv = 1;
  while ((1ll << lv) < 2ll * (long long)(M > 0 ? M : 1)) ++lv;
  return lv + 1;
}

// Wave-uniform prefix of lane_nearest: walks down from the root while only ONE child box is within the
// bound of any lane (boxes through scalar loads), and returns the node the per-lane walks can start at -
// with tight bounds (ICP after the first iteration) that is most of the way down, so the lanes only
// diverge over the last few levels.  Lanes with an infinite bound need every box: the walk stays at 1.
__device__ __forceinline__ int wave_common_root(const TreeView &t, float qx, float qy, float qz, bool active,
                                                float tau) {
  const cgrp_ptr box2 = (cgrp_ptr)t.box;
  const int M = t.M;
  int root = 1;
  while (root < M) {
    const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
    if (2 * root + 1 < M) {
      // two levels per step: the four grandchildren boxes are two adjacent 64-byte lines
      const f16v ba = box2[2 * root], bb = box2[2 * root + 1];
      const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{ba[0], ba[8]}, f2v{ba[1], ba[9]}, f2v{ba[2], ba[10]},
                                  f2v{ba[3], ba[11]}, f2v{ba[4], ba[12]}, f2v{ba[5], ba[13]});
      const f2v m23 = box_dist_pk(q2x, q2y, q2z, f2v{bb[0], bb[8]}, f2v{bb[1], bb[9]}, f2v{bb[2], bb[10]},
                                  f2v{bb[3], bb[11]}, f2v{bb[4], bb[12]}, f2v{bb[5], bb[13]});
      const bool n0 = __any(active && m01.x <= tau), n1 = __any(active && m01.y <= tau);
      const bool n2 = __any(active && m23.x <= tau), n3 = __any(active && m23.y <= tau);
      const int cnt = (int)n0 + (int)n1 + (int)n2 + (int)n3;
      if (cnt == 1) {
        root = 4 * root + (n1 ? 1 : n2 ? 2 : n3 ? 3 : 0);
        continue;
      }
      if (cnt == 2 && n0 == n1) root = 2 * root + (n2 ? 1 : 0);  // both inside one child: that child, and stop
      break;
    }
    const f16v b = box2[root];
    const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{b[0], b[8]}, f2v{b[1], b[9]}, f2v{b[2], b[10]},
                                f2v{b[3], b[11]}, f2v{b[4], b[12]}, f2v{b[5], b[13]});
    const float m0 = m01.x, m1 = m01.y;
    const bool n0 = __any(active && m0 <= tau), n1 = __any(active && m1 <= tau);
    if (n0 == n1) break;  // both needed (or neither: nothing left to find below)
    root = 2 * root + (n1 ? 1 : 0);
  }
  return root;
}

// The walk itself, shared by the 1-NN and the k-NN kernels.  leaf(leaf_number, tau) scans one bucket for
// the calling lane and returns the lane's new bound (<= tau); round() runs once per outer round with the
// whole wave converged (the k-NN kernel compacts its candidate buffer there).
// start: the node a lane's walk begins at - `root`, or (lane_start_at_leaf) a bucket below it whose path up to the
// root has its far[] entries filled in already.
#ifdef PCLMI_WALK_COUNT
__device__ unsigned int g_walk_count[4];  // [0] outer rounds, [1] descend steps, [2] leaf scans, [3] start-at-leaf steps: maxima
#endif
template <class LeafFn, class RoundFn>
__device__ __forceinline__ void lane_traverse(const TreeView &t, float qx, float qy, float qz, bool searchable,
                                              float tau, float (*far)[64], int lane, int root, LeafFn leaf,
                                              RoundFn round, int start = 0) {
  const int M = t.M;
  int node = start ? start : root;
  unsigned int trail = 0;  // bit l set: the node on the current path at level l is the second child visited
  bool active = searchable;
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#ifdef PCLMI_WALK_COUNT
  unsigned int n_round = 0, n_desc = 0;
#endif
  while (__any(active)) {
#ifdef PCLMI_WALK_COUNT
    ++n_round;
#endif
    // ---- descend towards the nearer child while the subtree can still hold something <= tau
    bool pruned = false;
    while (__any(active && !pruned && node < M)) {
#ifdef PCLMI_WALK_COUNT
      ++n_desc;
#endif
      if (active && !pruned && node < M) {
        if (2 * node + 1 < M) {
          // two levels per step (half as many dependent loads): the four grandchildren boxes; a child's
          // bound is the smaller of its two children's box distances
          // (a box is lo.xyz hi.xyz + two pad floats: 24 of its 32 bytes are fetched)
          const float *bf = t.box + (size_t)node * 32;
          const float4 a0 = *reinterpret_cast<const float4 *>(bf), a2 = *reinterpret_cast<const float4 *>(bf + 8);
          const float4 c0 = *reinterpret_cast<const float4 *>(bf + 16), c2 = *reinterpret_cast<const float4 *>(bf + 24);
          const float2 a1 = *reinterpret_cast<const float2 *>(bf + 4), a3 = *reinterpret_cast<const float2 *>(bf + 12);
          const float2 c1 = *reinterpret_cast<const float2 *>(bf + 20), c3 = *reinterpret_cast<const float2 *>(bf + 28);
          const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{a0.x, a2.x}, f2v{a0.y, a2.y}, f2v{a0.z, a2.z},
                                      f2v{a0.w, a2.w}, f2v{a1.x, a3.x}, f2v{a1.y, a3.y});
          const f2v m23 = box_dist_pk(q2x, q2y, q2z, f2v{c0.x, c2.x}, f2v{c0.y, c2.y}, f2v{c0.z, c2.z},
                                      f2v{c0.w, c2.w}, f2v{c1.x, c3.x}, f2v{c1.y, c3.y});
          const float lo01 = fminf(m01.x, m01.y), lo23 = fminf(m23.x, m23.y);
          const bool right = lo23 < lo01;                     // which child
          const float g0 = right ? m23.x : m01.x, g1 = right ? m23.y : m01.y;
          const bool gright = g1 < g0;                        // which grandchild inside it
          const float mn = gright ? g1 : g0;
          if (mn <= tau) {
            const int child = 2 * node + (right ? 1 : 0);
            node = 2 * child + (gright ? 1 : 0);
            const int lvl = 31 - __clz(node);
            far[lvl - 1][lane] = right ? lo01 : lo23;
            far[lvl][lane] = gright ? g0 : g1;
            trail &= ~(3u << (lvl - 1));
          } else {
            pruned = true;
          }
        } else {
          const float4 *b = reinterpret_cast<const float4 *>(t.box + (size_t)node * 16);
          const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
          const f2v m01 = box_dist_pk(q2x, q2y, q2z, f2v{b0.x, b2.x}, f2v{b0.y, b2.y}, f2v{b0.z, b2.z},
                                      f2v{b0.w, b2.w}, f2v{b1.x, b3.x}, f2v{b1.y, b3.y});
          const float m0 = m01.x, m1 = m01.y;
          const bool right = m1 < m0;
          const float mn = right ? m1 : m0, mf = right ? m0 : m1;
          if (mn <= tau) {
            node = 2 * node + (right ? 1 : 0);
            const int lvl = 31 - __clz(node);
            far[lvl][lane] = mf;
            trail &= ~(1u << lvl);
          } else {
            pruned = true;
          }
        }
      }
    }
    // ---- leaf bucket
    if (active && !pruned) tau = leaf(node - M, tau);
    round();
    // ---- back up to the deepest level whose far child is still within tau (non-strict: an equal
    //      distance with a lower index may hide there)
    if (active) {
      for (;;) {
        if (node == root) {
          active = false;
          break;
        }
        const int lvl = 31 - __clz(node);
        const unsigned int bit = 1u << lvl;
        if (!(trail & bit)) {
          trail |= bit;
          if (far[lvl][lane] <= tau) {
            node ^= 1;
            break;
          }
        }
        node >>= 1;
      }
    }
  }
#ifdef PCLMI_WALK_COUNT
  if (lane == 0) {
    atomicMax(&g_walk_count[0], n_round);
    atomicMax(&g_walk_count[1], n_desc);
    atomicAdd(&g_walk_count[2], n_round);
    atomicAdd(&g_walk_count[3], 1u);
  }
#endif
}

// Bottom-up start of a per-lane walk: the lane already knows a bucket that very likely holds its answer (ICP: the
// bucket of the previous iteration's neighbour).  Instead of walking down to it from `root` - one DEPENDENT load per
// two levels, which is what a converged ICP iteration spends its time on - the walk starts AT that bucket and the
// far[] entry of every level on the way up (the box distance of the path's sibling at that level, what the descent
// would have left behind) is computed from loads whose addresses are all known up front: four levels in flight at a
// time.  leaf_node must lie under `root`.
__device__ __forceinline__ void lane_start_at_leaf(const TreeView &t, float qx, float qy, float qz, bool use,
                                                   int leaf_node, int root, float (*far)[64], int lane) {
  int n = use ? leaf_node : root;
  while (__any(n > root)) {
    int id[4];
    float4 a[4];
    float2 b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      id[u] = n > root ? n : 0;  // 0: nothing left at this step for this lane
      n = n > root ? n >> 1 : n;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // unconditional loads (a finished lane re-reads the root's box): under `if (id[u])` each load becomes a branch
      // with its own s_waitcnt and the four levels are fetched one after the other
      const float *bp = t.box + (size_t)(id[u] ? (id[u] ^ 1) : 1) * 8;  // the sibling's box
      a[u] = *reinterpret_cast<const float4 *>(bp);
      b[u] = *reinterpret_cast<const float2 *>(bp + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (id[u]) far[31 - __clz(id[u])][lane] = box_dist(qx, qy, qz, a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y);
    }
  }
}

// start_leaf_node: heap id of a bucket to start the walk at (see lane_start_at_leaf), 0: start at `root`
__device__ __forceinline__ unsigned long long lane_nearest(const TreeView &t, float qx, float qy, float qz,
                                                           bool searchable, float tau0, float (*far)[64], int lane,
                                                           int root, int &best_slot, int start_leaf_node = 0) {
  // root: node the walk starts (and ends) at - 1, or wave_common_root(); best_slot: bucket slot of the result
  unsigned long long best = make_key(3.402823466e+38f, 0x7fffffff);
  int bslot = 0;
  const f2v q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
  const bool up = searchable && start_leaf_node > root;
  if (__any(up)) lane_start_at_leaf(t, qx, qy, qz, up, start_leaf_node, root, far, lane);
  lane_traverse(
      t, qx, qy, qz, searchable, tau0, far, lane, root,
      [&](int leafno, float tau) {  // all 16 slots (padded slots are at distance +inf)
        const float4 *rec = reinterpret_cast<const float4 *>(t.leaf + (size_t)leafno * (4 * L));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 X = rec[g * 4 + 0], Y = rec[g * 4 + 1], Z = rec[g * 4 + 2], I = rec[g * 4 + 3];
          const f2v d01 = sq3_pk(q2x - f2v{X.x, X.y}, q2y - f2v{Y.x, Y.y}, q2z - f2v{Z.x, Z.y});
          const f2v d23 = sq3_pk(q2x - f2v{X.z, X.w}, q2y - f2v{Y.z, Y.w}, q2z - f2v{Z.z, Z.w});
          const unsigned long long k0 = make_key(d01.x, __float_as_int(I.x));
          const unsigned long long k1 = make_key(d01.y, __float_as_int(I.y));
          const unsigned long long k2 = make_key(d23.x, __float_as_int(I.z));
          const unsigned long long k3 = make_key(d23.y, __float_as_int(I.w));
          const bool a1 = k1 < k0, b1 = k3 < k2;
          const unsigned long long ka = a1 ? k1 : k0, kb = b1 ? k3 : k2;
          const int pa = a1 ? 1 : 0, pb = b1 ? 3 : 2;
          const bool c1 = kb < ka;
          const unsigned long long kc = c1 ? kb : ka;
          const int pc = c1 ? pb : pa;
          if (kc < best) {
            best = kc;
            bslot = leafno * L + g * 4 + pc;
          }
        }
        return fminf(tau, __uint_as_float((unsigned int)(best >> 32)));
      },
      [] {}, up ? start_leaf_node : 0);
  best_slot = bslot;
  return best;
}

}  // namespace dev
}  // namespace pclmi
#endif
