// plane_kernels.hip - the point-to-plane pieces of the registration path on gfx950:
//   * TransformationEstimationPointToPlaneLLS::estimateRigidTransformation
//     (registration/include/pcl/registration/impl/transformation_estimation_point_to_plane_lls.hpp:141-245):
//     the 21 + 6 sums of the 6x6 normal equations reduced on the GPU in fp64 (wave shuffles -> LDS row per
//     wave -> per-block partial -> fixed-order pass), the 6x6 solve and constructTransformationMatrix
//     (:118-139) on the host;
//   * CorrespondenceEstimationNormalShooting::determineCorrespondences
//     (impl/correspondence_estimation_normal_shooting.hpp:60-165): one batched k-NN launch, then one lane per
//     source point picks the neighbour closest to the line along the source normal.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pclmi_internal.h"

namespace pclmi {
namespace {

constexpr int NP = 27;  // 21 upper-triangle entries of ATA (row-major) + 6 of ATb
constexpr int P_WPB = 4;

__device__ __forceinline__ bool fin3(float4 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// src / tgt / tn (/ sn) are already paired row by row
__global__ __launch_bounds__(64 * P_WPB) void p2plane_reduce_kernel(const float4 *__restrict__ src,
                                                                     const float4 *__restrict__ sn,
                                                                     const float4 *__restrict__ tgt,
                                                                     const float4 *__restrict__ tn, int n,
                                                                     double *__restrict__ partials) {
  __shared__ double lds[P_WPB][NP];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  double acc[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) acc[j] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 s = src[i], d = tgt[i], m = tn[i];
    if (!fin3(s) || !fin3(d) || !fin3(m) || (sn && !fin3(sn[i]))) continue;  // :156-172
    // fp32 expressions widened on assignment, as the reference writes them (:186-188, :214)
    const double a = (double)__fsub_rn(__fmul_rn(m.z, s.y), __fmul_rn(m.y, s.z));
    const double b = (double)__fsub_rn(__fmul_rn(m.x, s.z), __fmul_rn(m.z, s.x));
    const double c = (double)__fsub_rn(__fmul_rn(m.y, s.x), __fmul_rn(m.x, s.y));
    const double nx = m.x, ny = m.y, nz = m.z;
    float df = __fadd_rn(__fadd_rn(__fmul_rn(m.x, d.x), __fmul_rn(m.y, d.y)), __fmul_rn(m.z, d.z));
    df = __fsub_rn(__fsub_rn(__fsub_rn(df, __fmul_rn(m.x, s.x)), __fmul_rn(m.y, s.y)), __fmul_rn(m.z, s.z));
    const double dd = df;
    const double v[6] = {a, b, c, nx, ny, nz};
    int o = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int q = r; q < 6; ++q) acc[o++] += v[r] * v[q];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] += v[r] * dd;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    double x = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0) lds[wib][j] = x;
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < P_WPB; ++w) s += lds[w][threadIdx.x];
    partials[(size_t)blockIdx.x * NP + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(256) void sum_rows_kernel(const double *__restrict__ partials, int nrows, int nv,
                                                       double *__restrict__ out) {
  __shared__ double sh[256];
  const int j = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) s += partials[(size_t)r * nv + j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[j] = sh[0];
}

// one lane per source point: among its k neighbours (ascending distance) the first minimum of |N x V|^2
__global__ __launch_bounds__(256) void normal_shoot_kernel(const float4 *__restrict__ src,
                                                           const float4 *__restrict__ sn,
                                                           const float4 *__restrict__ cloud,
                                                           const int *__restrict__ idx, const float *__restrict__ d2,
                                                           int n, int k, double max_distance,
                                                           int *__restrict__ match, float *__restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = src[i], nr = sn[i];
  int best = -1;
  float best_d2 = 0.f;
  if (fin3(p)) {
    float min_dist = 3.402823466e+38f;
    int min_j = 0;
    const double N0 = nr.x, N1 = nr.y, N2 = nr.z;
    for (int j = 0; j < k; ++j) {
      const int t = idx[(size_t)i * k + j];
      if (t < 0) break;
      const float4 q = cloud[t];
      const double V0 = (double)__fsub_rn(q.x, p.x), V1 = (double)__fsub_rn(q.y, p.y), V2 = (double)__fsub_rn(q.z, p.z);
      const double C0 = N1 * V2 - N2 * V1, C1 = N2 * V0 - N0 * V2, C2 = N0 * V1 - N1 * V0;
      const float dcur = (float)((C0 * C0 + C1 * C1) + C2 * C2);
      if (dcur < min_dist) {
        min_dist = dcur;
        min_j = j;
      }
    }
    if (!((double)min_dist > max_distance) && idx[(size_t)i * k] >= 0) {  // :111-112
      best = idx[(size_t)i * k + min_j];
      best_d2 = d2[(size_t)i * k + min_j];
    }
  }
  match[i] = best;
  dist[i] = best_d2;
}

// ---- CorrespondenceEstimationBackProjection --------------------------------------------------------------
// The reference loop (impl/correspondence_estimation_backprojection.hpp:58-107) weighs neighbour j by
// nn_dists[min_index] * (2 - cos^2), where min_index is the running arg-min - and is NOT reset between source
// points (:71), so point i starts from the arg-min point i-1 ended with.  That serial chain carries one small
// integer, so it parallelises as a scan over function composition: every point gets the map
// "min_index on entry -> min_index on exit" (k entries), an exclusive scan composes the maps in source order,
// and entry 0 of the scanned map is the state each point really starts from.
struct BpMap {
  unsigned long long w[4];  // 32 entries of 8 bits
};
__host__ __device__ inline unsigned bp_get(const BpMap &a, unsigned i) {
  const unsigned long long x = i < 8 ? a.w[0] : i < 16 ? a.w[1] : i < 24 ? a.w[2] : a.w[3];
  return (unsigned)(x >> ((i & 7u) * 8u)) & 0xffu;
}
__host__ __device__ inline void bp_set(BpMap &a, unsigned i, unsigned v) {
  const unsigned long long m = (unsigned long long)(v & 0xffu) << ((i & 7u) * 8u);
  if (i < 8) a.w[0] |= m;
  else if (i < 16) a.w[1] |= m;
  else if (i < 24) a.w[2] |= m;
  else a.w[3] |= m;
}
__host__ __device__ inline BpMap bp_identity() {
  BpMap r;
  r.w[0] = 0x0706050403020100ull;
  r.w[1] = 0x0f0e0d0c0b0a0908ull;
  r.w[2] = 0x1716151413121110ull;
  r.w[3] = 0x1f1e1d1c1b1a1918ull;
  return r;
}
struct BpCompose {  // (a then b)
  __host__ __device__ BpMap operator()(const BpMap &a, const BpMap &b) const {
    BpMap r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0ull;
    for (unsigned s = 0; s < 32; ++s) bp_set(r, s, bp_get(b, bp_get(a, s)));
    return r;
  }
};

// weight[i][j] = 2 - cos^2 of the angle between source normal i and the normal of its j-th neighbour (:80-83)
__global__ __launch_bounds__(256) void bp_weight_kernel(const float4 *__restrict__ sn, const float4 *__restrict__ tn,
                                                        const int *__restrict__ idx, size_t total, int k,
                                                        float *__restrict__ weight) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int t = idx[e];
  float w = __builtin_nanf("");
  if (t >= 0) {
    const float4 a = sn[e / (size_t)k], b = tn[t];
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)), __fmul_rn(a.z, b.z));
    w = __fsub_rn(2.0f, __fmul_rn(c, c));
  }
  weight[e] = w;
}

// the reference's inner loop for one point, entered with min_index = s
__device__ __forceinline__ int bp_walk(const float *__restrict__ d2, const float *__restrict__ w, int k, int s,
                                       float &min_dist) {
  min_dist = 3.402823466e+38f;
  int mi = s;
  for (int j = 0; j < k; ++j) {
    const float dist = __fmul_rn(d2[mi], w[j]);
    if (dist < min_dist) {
      min_dist = dist;
      mi = j;
    }
  }
  return mi;
}

__global__ __launch_bounds__(256) void bp_map_kernel(const int *__restrict__ idx, const float *__restrict__ d2,
                                                     const float *__restrict__ weight, int n, int k,
                                                     BpMap *__restrict__ maps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  BpMap m = bp_identity();
  if (idx[(size_t)i * k] >= 0) {  // searchable point (others leave the state alone)
    m.w[0] = m.w[1] = m.w[2] = m.w[3] = 0ull;
    float md;
    for (int s = 0; s < 32; ++s)
      bp_set(m, s, s < k ? (unsigned)bp_walk(d2 + (size_t)i * k, weight + (size_t)i * k, k, s, md) : (unsigned)s);
  }
  maps[i] = m;
}

__global__ __launch_bounds__(256) void bp_pick_kernel(const int *__restrict__ idx, const float *__restrict__ d2,
                                                      const float *__restrict__ weight,
                                                      const BpMap *__restrict__ entry, int n, int k,
                                                      double max_distance, int *__restrict__ match,
                                                      float *__restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = -1;
  float best_d2 = 0.f;
  if (idx[(size_t)i * k] >= 0) {
    float md;
    const int mi = bp_walk(d2 + (size_t)i * k, weight + (size_t)i * k, k, (int)bp_get(entry[i], 0), md);
    if (!((double)md > max_distance)) {  // :91-92
      best = idx[(size_t)i * k + mi];
      best_d2 = d2[(size_t)i * k + mi];
    }
  }
  match[i] = best;
  dist[i] = best_d2;
}

// host rows (strided, optionally through an index list) -> device float4
int upload_rows(const float *h, size_t n_rows, size_t stride_bytes, const int32_t *rows, size_t n, DevBuf<float4> &out) {
  std::vector<float4> tmp(n ? n : 1);
  const size_t sf = stride_bytes / 4;
  for (size_t i = 0; i < n; ++i) {
    const size_t r = rows ? (size_t)rows[i] : i;
    if (r >= n_rows) {
      set_error("index %zu out of range (%zu rows)", r, n_rows);
      return PCLMI_ERR_INVALID;
    }
    const float *p = h + r * sf;
    tmp[i] = make_float4(p[0], p[1], p[2], 1.f);
  }
  RC(out.alloc(n));
  if (n) PCLMI_HIP(hipMemcpy(out.p, tmp.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  return PCLMI_OK;
}

}  // namespace
}  // namespace pclmi

using namespace pclmi;

extern "C" {

int pclmi_estimate_point_to_plane_lls(const float *src, size_t n_src, size_t src_stride_bytes,
                                      const float *src_normals, size_t sn_stride_bytes, const int32_t *src_idx,
                                      const float *tgt, size_t n_tgt, size_t tgt_stride_bytes,
                                      const float *tgt_normals, size_t tn_stride_bytes, const int32_t *tgt_idx,
                                      size_t n, float T[16]) {
  PCLMI_CHECK(src && tgt && tgt_normals && T, PCLMI_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  auto ok_stride = [](size_t s) { return s >= 12 && s % 4 == 0; };
  PCLMI_CHECK(ok_stride(src_stride_bytes) && ok_stride(tgt_stride_bytes) && ok_stride(tn_stride_bytes) &&
                  (!src_normals || ok_stride(sn_stride_bytes)),
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  PCLMI_CHECK(n > 0, PCLMI_ERR_NOT_ENOUGH, "no point pairs");
  PCLMI_CHECK(src_idx || n <= n_src, PCLMI_ERR_INVALID, "n exceeds the source cloud");
  PCLMI_CHECK(tgt_idx || n <= n_tgt, PCLMI_ERR_INVALID, "n exceeds the target cloud");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 pairs");
  DevBuf<float4> ds, dsn, dt, dtn;
  RC(upload_rows(src, n_src, src_stride_bytes, src_idx, n, ds));
  if (src_normals) RC(upload_rows(src_normals, n_src, sn_stride_bytes, src_idx, n, dsn));
  RC(upload_rows(tgt, n_tgt, tgt_stride_bytes, tgt_idx, n, dt));
  RC(upload_rows(tgt_normals, n_tgt, tn_stride_bytes, tgt_idx, n, dtn));
  unsigned blocks = (unsigned)((n + 64 * P_WPB - 1) / (64 * P_WPB));
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  DevBuf<double> partials, sums;
  RC(partials.alloc((size_t)blocks * NP));
  RC(sums.alloc(NP));
  hipLaunchKernelGGL(p2plane_reduce_kernel, dim3(blocks), dim3(64 * P_WPB), 0, nullptr, ds.p,
                     src_normals ? dsn.p : (const float4 *)nullptr, dt.p, dtn.p, (int)n, partials.p);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(NP), dim3(256), 0, nullptr, partials.p, (int)blocks, NP, sums.p);
  double h[NP];
  PCLMI_HIP(hipMemcpy(h, sums.p, sizeof(h), hipMemcpyDeviceToHost));
  return solve_point_to_plane(h, T);
}

int pclmi_correspondences_normal_shooting(const pclmi_tree *target, const float *src, size_t n_src,
                                          size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                                          const int32_t *src_indices, size_t n_indices, int k, double max_distance,
                                          pclmi_correspondence *out, size_t *n_out) {
  PCLMI_CHECK(target && n_out, PCLMI_ERR_INVALID, "NULL argument");
  *n_out = 0;
  PCLMI_CHECK(n_src == 0 || (src && src_normals && out), PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0 && sn_stride_bytes >= 12 && sn_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(k > 0, PCLMI_ERR_INVALID, "k must be positive (got %d)", k);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  const size_t n = src_indices ? n_indices : n_src;
  if (n == 0 || target->n == 0) return PCLMI_OK;
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 source points");
  PCLMI_HIP(hipSetDevice(target->device));
  DevBuf<float4> ds, dn;
  RC(upload_rows(src, n_src, stride_bytes, src_indices, n, ds));
  RC(upload_rows(src_normals, n_src, sn_stride_bytes, src_indices, n, dn));
  const int k_eff = (size_t)k > target->n ? (int)target->n : k;
  DevBuf<int> idx, match;
  DevBuf<float> d2, dist;
  RC(idx.alloc(n * (size_t)k_eff));
  RC(d2.alloc(n * (size_t)k_eff));
  RC(match.alloc(n));
  RC(dist.alloc(n));
  RC(pclmi_knn_dev(target, (const float *)ds.p, n, 16, k_eff, idx.p, d2.p, nullptr));
  hipLaunchKernelGGL(normal_shoot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, target->stream, ds.p, dn.p,
                     (const float4 *)target->d_cloud, idx.p, d2.p, (int)n, k_eff, max_distance, match.p, dist.p);
  PCLMI_HIP(hipGetLastError());
  PCLMI_HIP(hipStreamSynchronize(target->stream));
  std::vector<int> hm(n);
  std::vector<float> hd(n);
  PCLMI_HIP(hipMemcpy(hm.data(), match.p, n * 4, hipMemcpyDeviceToHost));
  PCLMI_HIP(hipMemcpy(hd.data(), dist.p, n * 4, hipMemcpyDeviceToHost));
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    if (hm[i] < 0) continue;
    out[c].index_query = src_indices ? src_indices[i] : (int32_t)i;
    out[c].index_match = hm[i];
    out[c].distance = hd[i];
    ++c;
  }
  *n_out = c;
  return PCLMI_OK;
}

int pclmi_correspondences_back_projection(const pclmi_tree *target, const float *src, size_t n_src,
                                          size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                                          const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes,
                                          const int32_t *src_indices, size_t n_indices, int k, double max_distance,
                                          pclmi_correspondence *out, size_t *n_out) {
  PCLMI_CHECK(target && n_out, PCLMI_ERR_INVALID, "NULL argument");
  *n_out = 0;
  PCLMI_CHECK(n_src == 0 || (src && src_normals && out), PCLMI_ERR_INVALID, "NULL argument");
  PCLMI_CHECK(n_tgt == 0 || tgt_normals, PCLMI_ERR_INVALID, "tgt_normals is NULL");
  auto ok_stride = [](size_t s) { return s >= 12 && s % 4 == 0; };
  PCLMI_CHECK(ok_stride(stride_bytes) && ok_stride(sn_stride_bytes) && ok_stride(tn_stride_bytes), PCLMI_ERR_INVALID,
              "bad stride");
  PCLMI_CHECK(k > 0, PCLMI_ERR_INVALID, "k must be positive (got %d)", k);
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  const size_t n = src_indices ? n_indices : n_src;
  if (n == 0 || target->n == 0) return PCLMI_OK;
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 source points");
  // the tree hands out rows of the cloud it was built on: the normals must cover every one of them
  PCLMI_CHECK(n_tgt >= target->rows, PCLMI_ERR_INVALID, "%zu target normals for a target cloud of %zu rows", n_tgt,
              (size_t)target->rows);
  const int k_eff = (size_t)k > target->n ? (int)target->n : k;
  PCLMI_CHECK(k_eff <= 32, PCLMI_ERR_UNSUPPORTED, "back projection supports k <= 32 (got %d)", k_eff);
  PCLMI_HIP(hipSetDevice(target->device));
  hipStream_t s = target->stream;
  DevBuf<float4> ds, dn, dtn;
  RC(upload_rows(src, n_src, stride_bytes, src_indices, n, ds));
  RC(upload_rows(src_normals, n_src, sn_stride_bytes, src_indices, n, dn));
  RC(upload_rows(tgt_normals, n_tgt, tn_stride_bytes, nullptr, n_tgt, dtn));
  DevBuf<int> idx, match;
  DevBuf<float> d2, weight, dist;
  DevBuf<BpMap> maps, entry;
  const size_t total = n * (size_t)k_eff;
  RC(idx.alloc(total));
  RC(d2.alloc(total));
  RC(weight.alloc(total));
  RC(match.alloc(n));
  RC(dist.alloc(n));
  RC(maps.alloc(n));
  RC(entry.alloc(n));
  RC(pclmi_knn_dev(target, (const float *)ds.p, n, 16, k_eff, idx.p, d2.p, nullptr));
  hipLaunchKernelGGL(bp_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dn.p, dtn.p, idx.p, total,
                     k_eff, weight.p);
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(bp_map_kernel, dim3(grid), dim3(256), 0, s, idx.p, d2.p, weight.p, (int)n, k_eff, maps.p);
  size_t tmp_bytes = 0;
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveScan(nullptr, tmp_bytes, maps.p, entry.p, BpCompose(), bp_identity(), (int)n, s));
  DevBuf<unsigned char> tmp;
  RC(tmp.alloc(tmp_bytes ? tmp_bytes : 1));
  PCLMI_HIP(hipcub::DeviceScan::ExclusiveScan(tmp.p, tmp_bytes, maps.p, entry.p, BpCompose(), bp_identity(), (int)n, s));
  hipLaunchKernelGGL(bp_pick_kernel, dim3(grid), dim3(256), 0, s, idx.p, d2.p, weight.p, entry.p, (int)n, k_eff,
                     max_distance, match.p, dist.p);
  PCLMI_HIP(hipGetLastError());
  PCLMI_HIP(hipStreamSynchronize(s));
  std::vector<int> hm(n);
  std::vector<float> hd(n);
  PCLMI_HIP(hipMemcpy(hm.data(), match.p, n * 4, hipMemcpyDeviceToHost));
  PCLMI_HIP(hipMemcpy(hd.data(), dist.p, n * 4, hipMemcpyDeviceToHost));
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    if (hm[i] < 0) continue;
    out[c].index_query = src_indices ? src_indices[i] : (int32_t)i;
    out[c].index_match = hm[i];
    out[c].distance = hd[i];
    ++c;
  }
  *n_out = c;
  return PCLMI_OK;
}

}  // extern "C"
