// tree_build_dev.hip - construction of the left-balanced bucket kd-tree ON the GPU (gfx950).
//
// Replaces the FLANN KDTreeSingleIndex build behind pcl::KdTreeFLANN::setInputCloud
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:72-74) - and, since the reference rebuilds the target tree on
// every determineCorrespondences (registration/include/pcl/registration/impl/correspondence_estimation.hpp:47-63),
// a step that IS on the registration path.  The tree is the one tree_build.cpp builds on the host, node for node and
// bucket for bucket (same split dimension, same split position, same in-bucket order when the indexed rows are in
// ascending order) - search results never depend on the shape anyway - but no point ever visits the host:
//
//   1. the indexed rows with finite coordinates are compacted into packed points (flag, scan, scatter);
//   2. three radix sorts give the point order along x, y and z (ties in input order); a fourth list keeps the input
//      order itself;
//   3. level by level (all segments of a level at once): the tight box of a segment is read off the ends of its
//      three sorted runs - no reduction -, the widest extent picks the split dimension, the left-balanced heap shape
//      fixes how many points go left (the rule of tree_build.cpp), the first `left` points of the run in that
//      dimension are marked, and all four lists are STABLY partitioned inside every segment with one scan and one
//      scatter - so each run stays sorted and nothing is ever sorted again (the classic presorted kd-tree build);
//   4. the buckets (16 consecutive positions of the input-order list) are written as the leaf records and the
//      bucket-slot rows.
// log2(M) levels of ~5 launches each: a 1M-point target builds in a few ms, 10M in tens of ms.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>

#include "pclmi_internal.h"

namespace pclmi {
namespace {

constexpr int L_ = kLeafSize;

__device__ __forceinline__ bool fin3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// rows of the caller's cloud -> (x, y, z, 1): the copy ICP gathers target points from
__global__ void cloud_pack_kernel(const float *__restrict__ xyz, int stride_f, int n, float4 *__restrict__ cloud) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = xyz + (size_t)i * stride_f;
  cloud[i] = make_float4(p[0], p[1], p[2], 1.f);
}

// candidate a (row indices[a] or a): keep[a] = its coordinates are finite; *bad = an index outside the cloud
__global__ void keep_flags_kernel(const float *__restrict__ xyz, int stride_f, int n, const int *__restrict__ indices,
                                  int cand, int *__restrict__ keep, unsigned int *__restrict__ bad) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= cand) return;
  const int row = indices ? indices[a] : a;
  int k = 0;
  if (row < 0 || row >= n) {
    atomicOr(bad, 1u);
  } else {
    const float *p = xyz + (size_t)row * stride_f;
    k = fin3(p[0], p[1], p[2]) ? 1 : 0;
  }
  keep[a] = k;
}

// order-preserving compaction: pk[pos[a]] = (x, y, z, row bits)
__global__ void pack_kept_kernel(const float *__restrict__ xyz, int stride_f, const int *__restrict__ indices, int cand,
                                 const int *__restrict__ keep, const int *__restrict__ pos, float4 *__restrict__ pk) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= cand || !keep[a]) return;
  const int row = indices ? indices[a] : a;
  const float *p = xyz + (size_t)row * stride_f;
  pk[pos[a]] = make_float4(p[0], p[1], p[2], __int_as_float(row));
}

// radix-sortable image of a float, -0 == +0 (the host comparator's `<` does not tell them apart)
__device__ __forceinline__ unsigned int ordered_bits(float v) {
  const unsigned int u = __float_as_uint(v + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void sort_keys_kernel(const float4 *__restrict__ pk, int n, unsigned int *__restrict__ kx,
                                 unsigned int *__restrict__ ky, unsigned int *__restrict__ kz, int *__restrict__ ids,
                                 int *__restrict__ node_of_pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pk[i];
  kx[i] = ordered_bits(p.x);
  ky[i] = ordered_bits(p.y);
  kz[i] = ordered_bits(p.z);
  ids[i] = i;
  node_of_pos[i] = 1;
}

// leaf ids (M .. 2M-1) inside the subtree of heap id `node` (tree_build.cpp leaves_under)
__device__ __forceinline__ int leaves_under(long long node, int M) {
  long long lo = node, hi = node, cnt = 0;
  const long long first = M, last = 2ll * M - 1;
  while (lo <= last) {
    const long long a = lo > first ? lo : first, b = hi < last ? hi : last;
    if (a <= b) cnt += b - a + 1;
    lo = 2 * lo;
    hi = 2 * hi + 1;
  }
  return (int)cnt;
}

struct Lists {
  const int *l[4];  // point ids ordered by x, y, z and by input position, segment by segment
};
struct ListsOut {
  int *l[4];
};

// One thread per node of the level: its tight box from the ends of the three sorted runs; internal nodes also pick the
// split dimension and hand their range to their children.
__global__ void level_nodes_kernel(int first, int count, int M, const float4 *__restrict__ pk, Lists in,
                                   int *__restrict__ begin, int *__restrict__ end, int *__restrict__ dim,
                                   float *__restrict__ box) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const int node = first + j;
  const int a = begin[node], b = end[node];
  const float lo0 = pk[in.l[0][a]].x, hi0 = pk[in.l[0][b - 1]].x;
  const float lo1 = pk[in.l[1][a]].y, hi1 = pk[in.l[1][b - 1]].y;
  const float lo2 = pk[in.l[2][a]].z, hi2 = pk[in.l[2][b - 1]].z;
  float *bx = box + (size_t)node * 8;
  bx[0] = lo0; bx[1] = lo1; bx[2] = lo2;
  bx[3] = hi0; bx[4] = hi1; bx[5] = hi2;
  bx[6] = 0.f; bx[7] = 0.f;
  if (node >= M) return;  // a bucket
  int d = 0;
  float best = hi0 - lo0;
  if (hi1 - lo1 > best) { best = hi1 - lo1; d = 1; }
  if (hi2 - lo2 > best) { best = hi2 - lo2; d = 2; }
  dim[node] = d;
  // proportional split; every bucket keeps <= L points and at least one (tree_build.cpp)
  const int nl = leaves_under(2ll * node, M), nr = leaves_under(2ll * node + 1, M);
  const long long cnt = b - a;
  long long left = llround((double)cnt * nl / (double)(nl + nr));
  const long long max_left = min(cnt - (long long)nr, (long long)nl * L_);
  const long long min_left = max((long long)nl, cnt > (long long)nr * L_ ? cnt - (long long)nr * L_ : 0ll);
  left = min(max(left, min_left), max_left);
  begin[2 * node] = a;
  end[2 * node] = a + (int)left;
  begin[2 * node + 1] = a + (int)left;
  end[2 * node + 1] = b;
}

// One thread per position: which side of its segment's split does the point at this position of the split dimension's
// run go to (a finished bucket keeps everything "left": its partition is the identity), and which node owns the
// position afterwards.
__global__ void level_mark_kernel(int n, int M, const int *__restrict__ node_of_pos, int *__restrict__ node_next,
                                  const int *__restrict__ begin, const int *__restrict__ dim, Lists in,
                                  unsigned char *__restrict__ side) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = node_of_pos[i];
  if (node >= M) {
    side[in.l[0][i]] = 0;
    node_next[i] = node;
    return;
  }
  const int mid = begin[2 * node + 1];  // first position of the right child
  const int right = i >= mid ? 1 : 0;
  side[in.l[dim[node]][i]] = (unsigned char)right;
  node_next[i] = 2 * node + right;
}

// is_left flag of virtual position v = list * n + i (one scan serves the four lists)
struct LeftFlag {
  const unsigned char *side;
  const int *l0, *l1, *l2, *l3;
  int n;
  __host__ __device__ int operator()(int v) const {
    const int which = v / n, i = v - which * n;
    const int *l = which == 0 ? l0 : which == 1 ? l1 : which == 2 ? l2 : l3;
    return side[l[i]] ? 0 : 1;
  }
};

// stable partition of every segment of the four lists: P = exclusive prefix of the is_left flags
__global__ void level_scatter_kernel(int n, int M, const int *__restrict__ node_of_pos, const int *__restrict__ begin,
                                     const unsigned char *__restrict__ side, const int *__restrict__ P, Lists in,
                                     ListsOut out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = node_of_pos[i];
  const int a = begin[node];
  const int mid = node >= M ? 0 : begin[2 * node + 1];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int pt = in.l[w][i];
    const int lefts_before = P[(size_t)w * n + i] - P[(size_t)w * n + a];
    const int dst = side[pt] ? mid + ((i - a) - lefts_before) : a + lefts_before;
    out.l[w][dst] = pt;
  }
}

// buckets -> leaf records (per group of 4 points x[4] y[4] z[4] idx[4]) and bucket-slot rows (x, y, z, idx bits)
__global__ void leaves_fill_kernel(int M, const int *__restrict__ begin, const int *__restrict__ end,
                                   const int *__restrict__ by_input, const float4 *__restrict__ pk,
                                   float *__restrict__ leaf, float4 *__restrict__ slot) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * L_) return;
  const int lf = t / L_, j = t - lf * L_;
  const int a = begin[M + lf], cnt = end[M + lf] - a;
  float4 p = make_float4(kPadCoord, kPadCoord, kPadCoord, __int_as_float(-1));
  if (j < cnt) p = pk[by_input[a + j]];
  float *gp = leaf + (size_t)lf * 4 * L_ + (j >> 2) * 16 + (j & 3);
  gp[0] = p.x;
  gp[4] = p.y;
  gp[8] = p.z;
  gp[12] = p.w;
  slot[t] = p;
}

__global__ void root_copy_kernel(float *__restrict__ box) {
  if (threadIdx.x < 8) box[threadIdx.x] = box[8 + threadIdx.x];  // id 0 is unused: a stray read sees the root box
}

// sum of the buckets' box volumes (a statistic, not a result: see pclmi_tree::fill)
__global__ __launch_bounds__(256) void leaf_volume_kernel(const float *__restrict__ box, int M, double *__restrict__ out) {
  __shared__ double part[256];
  double v = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
    const float *b = box + (size_t)(M + i) * 8;
    v += (double)(b[3] - b[0]) * (double)(b[4] - b[1]) * (double)(b[5] - b[2]);
  }
  part[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// Builds t's device arrays (d_box, d_leaf, d_slot, d_cloud, M, n, bounding box) from a cloud in DEVICE memory.
// d_indices: nullable device list of the rows to index.  Synchronises the stream (sizes are needed on the host).
int build_device_tree(pclmi_tree *t, const float *d_xyz, size_t n_rows, size_t stride_f, const int32_t *d_indices,
                      size_t n_indices, hipStream_t s) {
  const size_t cand = d_indices ? n_indices : n_rows;
  t->rows = n_rows;
  t->n = 0;
  t->M = 0;
  if (n_rows) {
    PCLMI_HIP(hipMalloc(&t->d_cloud, n_rows * sizeof(float4)));
    hipLaunchKernelGGL(cloud_pack_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, d_xyz, (int)stride_f, (int)n_rows,
                       (float4 *)t->d_cloud);
  }
  if (cand == 0) {
    PCLMI_HIP(hipStreamSynchronize(s));
    return PCLMI_OK;
  }
  // ---- 1. finite rows -> packed points
  DevBuf<int> keep, pos;
  DevBuf<unsigned int> bad;
  DevBuf<unsigned char> tmp;
  RC(keep.alloc(cand + 1));
  RC(pos.alloc(cand + 1));
  RC(bad.alloc(1));
  PCLMI_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned int), s));
  PCLMI_HIP(hipMemsetAsync(keep.p + cand, 0, sizeof(int), s));
  hipLaunchKernelGGL(keep_flags_kernel, dim3(grid_for(cand)), dim3(256), 0, s, d_xyz, (int)stride_f, (int)n_rows,
                     d_indices, (int)cand, keep.p, bad.p);
  size_t tmp_bytes = 0;
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, keep.p, pos.p, (int)(cand + 1), s));
  RC(tmp.alloc(tmp_bytes + 256));
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, keep.p, pos.p, (int)(cand + 1), s));
  int n_fin = 0;
  unsigned int h_bad = 0;
  PCLMI_HIP(hipMemcpyAsync(&n_fin, pos.p + cand, sizeof(int), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(h_bad), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  PCLMI_CHECK(!h_bad, PCLMI_ERR_INVALID, "an entry of the index list is outside the cloud (%zu rows)", n_rows);
  if (n_fin == 0) return PCLMI_OK;
  const size_t n = (size_t)n_fin;
  const int M = (int)((n + L_ - 1) / L_);
  t->n = n;
  t->M = M;
  DevBuf<float4> pk;
  RC(pk.alloc(n));
  hipLaunchKernelGGL(pack_kept_kernel, dim3(grid_for(cand)), dim3(256), 0, s, d_xyz, (int)stride_f, d_indices, (int)cand,
                     keep.p, pos.p, pk.p);
  keep.reset();
  // ---- 2. the three sorted orders
  DevBuf<unsigned int> keys;  // kx ky kz | sorted keys (discarded)
  DevBuf<int> lists;          // 2 x 4 lists (double buffer)
  DevBuf<int> node_pos;       // 2 x n
  DevBuf<int> range;          // begin[2M] end[2M] dim[2M]
  DevBuf<int> prefix;         // 4n + 1
  DevBuf<unsigned char> side;
  RC(keys.alloc(4 * n));
  RC(lists.alloc(8 * n));
  RC(node_pos.alloc(2 * n));
  RC(range.alloc((size_t)6 * M));
  RC(prefix.alloc(4 * n + 1));
  RC(side.alloc(n));
  int *cur[4] = {lists.p, lists.p + n, lists.p + 2 * n, lists.p + 3 * n};
  int *alt[4] = {lists.p + 4 * n, lists.p + 5 * n, lists.p + 6 * n, lists.p + 7 * n};
  int *node_cur = node_pos.p, *node_alt = node_pos.p + n;
  int *begin = range.p, *end = range.p + 2 * (size_t)M, *dim = range.p + 4 * (size_t)M;
  // ids in input order: list 3 from the start, and the values of the three sorts
  hipLaunchKernelGGL(sort_keys_kernel, dim3(grid_for(n)), dim3(256), 0, s, (const float4 *)pk.p, (int)n, keys.p,
                     keys.p + n, keys.p + 2 * n, cur[3], node_cur);
  size_t sort_bytes = 0;
  PCLMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned int *)keys.p, keys.p + 3 * n,
                                               (const int *)cur[3], cur[0], (int)n, 0, 32, s));
  size_t scan_bytes = 0;
  {
    hipcub::CountingInputIterator<int> cnt_it(0);
    LeftFlag f{side.p, cur[0], cur[1], cur[2], cur[3], (int)n};
    hipcub::TransformInputIterator<int, LeftFlag, hipcub::CountingInputIterator<int>> it(cnt_it, f);
    PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, it, prefix.p, (int)(4 * n), s));
  }
  RC(tmp.alloc((sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 256));
  for (int d = 0; d < 3; ++d) {
    size_t bytes = sort_bytes;
    PCLMI_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, (const unsigned int *)(keys.p + (size_t)d * n),
                                                 keys.p + 3 * n, (const int *)cur[3], cur[d], (int)n, 0, 32, s));
  }
  // ---- 3. level by level
  PCLMI_HIP(hipMalloc(&t->d_box, (size_t)2 * M * 8 * sizeof(float)));
  {
    const int root_range[2] = {0, (int)n};
    PCLMI_HIP(hipMemcpyAsync(begin + 1, &root_range[0], sizeof(int), hipMemcpyHostToDevice, s));
    PCLMI_HIP(hipMemcpyAsync(end + 1, &root_range[1], sizeof(int), hipMemcpyHostToDevice, s));
    PCLMI_HIP(hipStreamSynchronize(s));  // root_range is a stack variable
  }
  for (long long first = 1; first <= 2ll * M - 1; first *= 2) {
    const long long last = std::min(2 * first - 1, 2ll * M - 1);
    const int count = (int)(last - first + 1);
    Lists in{{cur[0], cur[1], cur[2], cur[3]}};
    hipLaunchKernelGGL(level_nodes_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, s, (int)first, count, M,
                       (const float4 *)pk.p, in, begin, end, dim, t->d_box);
    if (first >= M) break;  // the level holds buckets only: nothing left to split
    hipLaunchKernelGGL(level_mark_kernel, dim3(grid_for(n)), dim3(256), 0, s, (int)n, M, (const int *)node_cur, node_alt,
                       (const int *)begin, (const int *)dim, in, side.p);
    {
      hipcub::CountingInputIterator<int> cnt_it(0);
      LeftFlag f{side.p, cur[0], cur[1], cur[2], cur[3], (int)n};
      hipcub::TransformInputIterator<int, LeftFlag, hipcub::CountingInputIterator<int>> it(cnt_it, f);
      size_t bytes = scan_bytes;
      PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, it, prefix.p, (int)(4 * n), s));
    }
    ListsOut out{{alt[0], alt[1], alt[2], alt[3]}};
    hipLaunchKernelGGL(level_scatter_kernel, dim3(grid_for(n)), dim3(256), 0, s, (int)n, M, (const int *)node_cur,
                       (const int *)begin, (const unsigned char *)side.p, (const int *)prefix.p, in, out);
    for (int w = 0; w < 4; ++w) std::swap(cur[w], alt[w]);
    std::swap(node_cur, node_alt);
  }
  // ---- 4. buckets
  PCLMI_HIP(hipMalloc(&t->d_leaf, (size_t)M * 4 * L_ * sizeof(float)));
  PCLMI_HIP(hipMalloc(&t->d_slot, (size_t)M * L_ * sizeof(float4)));
  hipLaunchKernelGGL(leaves_fill_kernel, dim3(grid_for((size_t)M * L_)), dim3(256), 0, s, M, (const int *)begin,
                     (const int *)end, (const int *)cur[3], (const float4 *)pk.p, t->d_leaf, (float4 *)t->d_slot);
  hipLaunchKernelGGL(root_copy_kernel, dim3(1), dim3(64), 0, s, t->d_box);
  DevBuf<double> vol;
  RC(vol.alloc(1));
  PCLMI_HIP(hipMemsetAsync(vol.p, 0, sizeof(double), s));
  hipLaunchKernelGGL(leaf_volume_kernel, dim3((unsigned)((M + 1023) / 1024 < 256 ? (M + 1023) / 1024 : 256)), dim3(256), 0,
                     s, (const float *)t->d_box, M, vol.p);
  PCLMI_HIP(hipGetLastError());
  float root[8];
  double h_vol = 0.0;
  PCLMI_HIP(hipMemcpyAsync(root, t->d_box + 8, sizeof(root), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipMemcpyAsync(&h_vol, vol.p, sizeof(double), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  for (int d = 0; d < 3; ++d) {
    t->bb_lo[d] = root[d];
    t->bb_hi[d] = root[3 + d];
  }
  const double bbv = (double)(root[3] - root[0]) * (double)(root[4] - root[1]) * (double)(root[5] - root[2]);
  t->fill = bbv > 0.0 ? (float)(h_vol / bbv) : 0.f;
  return PCLMI_OK;
}

}  // namespace pclmi
