// reject_kernels.hip - the modular registration pipeline's selection stages on gfx950 (MI355X) + their C ABI.
//
// Replaces CorrespondenceEstimation::determineCorrespondences' keep test
// (registration/include/pcl/registration/impl/correspondence_estimation.hpp:83-141) and the CorrespondenceRejector*
// family (impl/correspondence_rejection_distance.hpp:42-69, _median_distance.hpp:44-69, _one_to_one.hpp:42-67,
// _trimmed.hpp:42-58, _var_trimmed.hpp:44-107).  pcl::Correspondence is a 12-byte AoS record; every stage is a
// selection over n records that keeps the input order:
//   distance / median / var-trimmed (by distance) / correspondence keep test:
//       ONE hand-written order-preserving compaction (predicate -> wave ballots -> LDS prefix over the tile ->
//       tile offsets from a scan of the tile counts -> scatter); the predicate is evaluated twice instead of
//       storing a flag array;
//   one-to-one : sort by (match, distance, query) [hipcub merge sort], keep the head of each match run (same compaction);
//   trimmed    : sort by (distance, query), keep the first floor(ratio * n);
//   median / var-trimmed thresholds: radix sort of the distances [hipcub], then a one-workgroup kernel (k-th element,
//       the FRMS arg-min) - only a few scalars ever cross PCIe.
// Every stage has a _dev form (records resident in HBM, work enqueued on the caller's stream, scratch from a pooled
// per-device workspace: no hipMalloc in the steady state); the host-pointer forms stage through the same pool.
// Sorting keys are total orders (ties broken by the query index; the reference's std::sort leaves ties unspecified -
// "parity unpinned" there, defined by the oracle).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "pclmi_internal.h"

namespace pclmi {
namespace {

typedef pclmi_correspondence Corr;

// ---- per-device scratch pool -------------------------------------------------------------------------------------
struct RejWs {
  int device = 0;
  hipStream_t own = nullptr;  // stream of the host-pointer forms
  void *buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[6] = {0, 0, 0, 0, 0, 0};
  int ensure(int which, size_t bytes) {
    if (bytes <= cap[which]) return PCLMI_OK;
    if (buf[which]) (void)hipFree(buf[which]);
    buf[which] = nullptr;
    cap[which] = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    PCLMI_HIP(hipMalloc(&buf[which], want));
    cap[which] = want;
    return PCLMI_OK;
  }
};
std::mutex g_rej_mu;
std::vector<RejWs *> g_rej_pool;

RejWs *acquire_rej() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> g(g_rej_mu);
    for (size_t i = 0; i < g_rej_pool.size(); ++i)
      if (g_rej_pool[i]->device == dev) {
        RejWs *w = g_rej_pool[i];
        g_rej_pool.erase(g_rej_pool.begin() + (long)i);
        return w;
      }
  }
  RejWs *w = new RejWs();
  w->device = dev;
  if (hipStreamCreateWithFlags(&w->own, hipStreamNonBlocking) != hipSuccess) {
    (void)hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    delete w;
    return nullptr;
  }
  return w;
}
struct RejGuard {
  RejWs *w;
  RejGuard() : w(acquire_rej()) {}
  ~RejGuard() {
    if (!w) return;
    std::lock_guard<std::mutex> g(g_rej_mu);
    g_rej_pool.push_back(w);
  }
};
enum { B_IN = 0, B_OUT = 1, B_A = 2, B_B = 3, B_TMP = 4, B_SMALL = 5 };

// ---- record sources: what the compaction reads and the predicate it applies -----------------------------------------
struct SrcDistLess {  // original.distance < limit (correspondence_rejection_distance.hpp:61, strict)
  const Corr *c;
  float limit;
  __device__ bool get(int i, Corr &r) const {
    r = c[i];
    return r.distance < limit;
  }
};
// (double)((ns.x * nt.x + ns.y * nt.y) + ns.z * nt.z) > threshold, the normals looked up through the pair's indices
// (DataContainer::getCorrespondenceScoreFromNormals, correspondence_rejection.h:254-262; _surface_normal.hpp:59-61)
struct SrcNormalDot {
  const Corr *c;
  const float *sn, *tn;
  int sn_stride_f, tn_stride_f, n_src, n_tgt;
  double threshold;
  __device__ bool get(int i, Corr &r) const {
    r = c[i];
    if ((unsigned int)r.index_query >= (unsigned int)n_src || (unsigned int)r.index_match >= (unsigned int)n_tgt)
      return false;  // (the reference would read past its clouds)
    const float *a = sn + (size_t)r.index_query * sn_stride_f, *b = tn + (size_t)r.index_match * tn_stride_f;
    const float dot = __fadd_rn(__fadd_rn(__fmul_rn(a[0], b[0]), __fmul_rn(a[1], b[1])), __fmul_rn(a[2], b[2]));
    return (double)dot > threshold;
  }
};
struct SrcDistLessEqD {  // (double) distance <= limit (correspondence_rejection_median_distance.hpp:64)
  const Corr *c;
  const double *limit;  // device scalar
  __device__ bool get(int i, Corr &r) const {
    r = c[i];
    return (double)r.distance <= *limit;
  }
};
struct SrcDistLessD {  // (double) distance < limit (correspondence_rejection_var_trimmed.hpp:72)
  const Corr *c;
  const double *limit;
  __device__ bool get(int i, Corr &r) const {
    r = c[i];
    return (double)r.distance < *limit;
  }
};
struct SrcRunHead {  // head of each run of equal non-negative match indices in a sorted list (_one_to_one.hpp:55-65)
  const Corr *c;
  __device__ bool get(int i, Corr &r) const {
    r = c[i];
    return r.index_match >= 0 && (i == 0 || c[i - 1].index_match != r.index_match);
  }
};
struct SrcNearest {  // determineCorrespondences: pair (query row, nearest target row), kept iff d2 <= max2 (:108)
  const int *nn;
  const float *d2;
  const int *rows;  // nullable: PCLBase indices_
  double max2;
  __device__ bool get(int i, Corr &r) const {
    r.index_query = rows ? rows[i] : i;
    r.index_match = nn[i];
    r.distance = d2[i];
    return r.index_match >= 0 && !((double)r.distance > max2);
  }
};

constexpr int CP_TPB = 256, CP_ROUNDS = 8, CP_TILE = CP_TPB * CP_ROUNDS;

// pass A: kept records per tile
template <class Src>
__global__ __launch_bounds__(CP_TPB) void compact_count_kernel(Src src, int n, unsigned int *__restrict__ tile_count) {
  __shared__ unsigned int wsum[CP_TPB / 64];
  const int base = blockIdx.x * CP_TILE;
  unsigned int c = 0;
#pragma unroll
  for (int r = 0; r < CP_ROUNDS; ++r) {
    const int i = base + r * CP_TPB + threadIdx.x;
    Corr rec;
    const bool keep = i < n && src.get(i, rec);
    c += (unsigned int)__popcll(__ballot(keep));  // wave-uniform
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
#pragma unroll
    for (int w = 0; w < CP_TPB / 64; ++w) t += wsum[w];
    tile_count[blockIdx.x] = t;
  }
}

// exclusive scan of the tile counts by one workgroup (any number of tiles: chunks of 1024 with a running carry);
// total -> *n_out
__global__ __launch_bounds__(1024) void tile_scan_kernel(unsigned int *__restrict__ tile_count, int ntiles,
                                                         unsigned int *__restrict__ n_out) {
  __shared__ unsigned int sh[1024];
  __shared__ unsigned int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int a = 0; a < ntiles; a += 1024) {
    const int i = a + threadIdx.x;
    const unsigned int v = i < ntiles ? tile_count[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned int add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < ntiles) tile_count[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = carry;
}

// pass B: the kept records to their places (input order kept)
template <class Src>
__global__ __launch_bounds__(CP_TPB) void compact_write_kernel(Src src, int n, const unsigned int *__restrict__ tile_off,
                                                               Corr *__restrict__ out) {
  __shared__ unsigned int cnt[CP_ROUNDS][CP_TPB / 64];
  const int base = blockIdx.x * CP_TILE;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  Corr rec[CP_ROUNDS];
  unsigned long long bal[CP_ROUNDS];
  bool keep[CP_ROUNDS];
#pragma unroll
  for (int r = 0; r < CP_ROUNDS; ++r) {
    const int i = base + r * CP_TPB + threadIdx.x;
    keep[r] = i < n && src.get(i, rec[r]);
    bal[r] = __ballot(keep[r]);
    if (lane == 0) cnt[r][wib] = (unsigned int)__popcll(bal[r]);
  }
  __syncthreads();
  // position = tile offset + kept records of earlier (round, wave) slots + kept lanes below this one
  unsigned int run = tile_off[blockIdx.x];
#pragma unroll
  for (int r = 0; r < CP_ROUNDS; ++r) {
#pragma unroll
    for (int w = 0; w < CP_TPB / 64; ++w) {
      if (w == wib && keep[r]) out[run + (unsigned int)__popcll(bal[r] & ((1ull << lane) - 1ull))] = rec[r];
      run += cnt[r][w];
    }
  }
}

template <class Src>
int compact(RejWs *w, Src src, size_t n, Corr *d_out, unsigned int *d_count, hipStream_t s) {
  const int ntiles = (int)((n + CP_TILE - 1) / CP_TILE);
  RC(w->ensure(B_TMP, (size_t)ntiles * sizeof(unsigned int)));
  unsigned int *tiles = (unsigned int *)w->buf[B_TMP];
  hipLaunchKernelGGL((compact_count_kernel<Src>), dim3(ntiles), dim3(CP_TPB), 0, s, src, (int)n, tiles);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, tiles, ntiles, d_count);
  hipLaunchKernelGGL((compact_write_kernel<Src>), dim3(ntiles), dim3(CP_TPB), 0, s, src, (int)n, (const unsigned int *)tiles,
                     d_out);
  PCLMI_HIP(hipGetLastError());
  return PCLMI_OK;
}

// small device results: [0] count (u32), [2..3] double a, [4..5] double b, [6..7] double c, [8] u32 cnt2
struct Small {
  unsigned int count, pad;
  double a, b, c;
  unsigned int cnt2, pad2;
};

int read_small(RejWs *w, Small *h, hipStream_t s) {
  PCLMI_HIP(hipMemcpyAsync(h, w->buf[B_SMALL], sizeof(Small), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  return PCLMI_OK;
}

__global__ void distances_kernel(const Corr *__restrict__ c, int n, float *__restrict__ d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = c[i].distance;
}

__global__ void kth_kernel(const float *__restrict__ sorted, int k, double factor, Small *__restrict__ sm) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    sm->a = (double)sorted[k];  // nth_element(n / 2) (_median_distance.hpp:60-61)
    sm->b = sm->a * factor;     // the threshold of :64
  }
}

// optimizeInlierRatio (_var_trimmed.hpp:82-107) on the ascending distances: lower_sum over the first min_el, then the
// first arg-min of FRMS_i = (1 / ratio_i^lambda)^2 * (1 / id_i) * (lower_sum + sorted[min_el + i]); then the trimmed
// distance sorted[int(n * factor)] and how many sorted distances lie strictly below it.
__global__ __launch_bounds__(1024) void frms_kernel(const float *__restrict__ sorted, int n, int min_el, int max_el,
                                                     double lambda, Small *__restrict__ sm) {
  __shared__ double shv[1024];
  __shared__ int shi[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < min_el; i += 1024) s += (double)sorted[i];
  shv[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) shv[t] += shv[t + o];
    __syncthreads();
  }
  const double lower_sum = shv[0];
  __syncthreads();
  double best = 0.0;
  int bi = -1;
  for (int i = t; i < max_el - min_el; i += 1024) {
    const double id = (double)(min_el + 1 + i);  // LinSpaced(max_el - min_el, min_el + 1, max_el): step 1
    const double ratio = id / (double)(unsigned int)n;
    const double inv = 1.0 / pow(ratio, lambda);
    const double frms = inv * inv * (1.0 / id) * (lower_sum + (double)sorted[min_el + i]);
    if (bi < 0 || frms < best) {  // ascending i per thread: the first minimum stays
      best = frms;
      bi = i;
    }
  }
  shv[t] = best;
  shi[t] = bi;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) {
      const int oi = shi[t + o];
      const double ov = shv[t + o];
      if (oi >= 0 && (shi[t] < 0 || ov < shv[t] || (ov == shv[t] && oi < shi[t]))) {
        shv[t] = ov;
        shi[t] = oi;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const int min_index = shi[0] < 0 ? 0 : shi[0];
    const float opt_ratio = (float)(min_index + min_el) / (float)(unsigned int)n;
    const double factor = (double)opt_ratio;
    long long pos = (long long)(int)((double)n * factor);
    if (pos >= n) pos = n - 1;
    const double trimmed = (double)sorted[pos];  // nth_element(int(n * factor)) (:64-65)
    // sorted[i] < trimmed: a prefix of the ascending array - its length by bisection
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((double)sorted[mid] < trimmed) lo = mid + 1;
      else hi = mid;
    }
    sm->a = factor;
    sm->b = trimmed;
    sm->cnt2 = (unsigned int)lo;
  }
}

struct ByMatchDistQuery {  // sortCorrespondencesByMatchIndexAndDistance (correspondence_sorting.h:109-120)
  __host__ __device__ bool operator()(const Corr &a, const Corr &b) const {
    if (a.index_match != b.index_match) return a.index_match < b.index_match;
    if (a.distance != b.distance) return a.distance < b.distance;
    return a.index_query < b.index_query;
  }
};
struct ByDistQuery {  // sortCorrespondencesByDistance (correspondence_sorting.h:79-86)
  __host__ __device__ bool operator()(const Corr &a, const Corr &b) const {
    if (a.distance != b.distance) return a.distance < b.distance;
    return a.index_query < b.index_query;
  }
};

// records -> a sorted copy in B_A (the caller's input is const)
template <typename Less>
int sorted_copy(RejWs *w, const Corr *d_in, size_t n, Less less, Corr **sorted, hipStream_t s) {
  RC(w->ensure(B_A, n * sizeof(Corr)));
  Corr *c = (Corr *)w->buf[B_A];
  PCLMI_HIP(hipMemcpyAsync(c, d_in, n * sizeof(Corr), hipMemcpyDeviceToDevice, s));
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceMergeSort::SortKeys(nullptr, bytes, c, (int)n, less, s));
  RC(w->ensure(B_TMP, bytes + 256));
  bytes = w->cap[B_TMP];
  PCLMI_HIP(hipcub::DeviceMergeSort::SortKeys(w->buf[B_TMP], bytes, c, (int)n, less, s));
  *sorted = c;
  return PCLMI_OK;
}

// ascending fp32 distances of the records, in B_B
int sorted_distances(RejWs *w, const Corr *d_in, size_t n, const float **sorted, hipStream_t s) {
  RC(w->ensure(B_A, n * sizeof(float)));
  RC(w->ensure(B_B, n * sizeof(float)));
  float *d = (float *)w->buf[B_A], *ds = (float *)w->buf[B_B];
  hipLaunchKernelGGL(distances_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (int)n, d);
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, d, ds, (int)n, 0, 32, s));
  RC(w->ensure(B_TMP, bytes + 256));
  bytes = w->cap[B_TMP];
  PCLMI_HIP(hipcub::DeviceRadixSort::SortKeys(w->buf[B_TMP], bytes, d, ds, (int)n, 0, 32, s));
  *sorted = ds;
  return PCLMI_OK;
}

int dev_checks(const Corr *in, size_t n, Corr *out, size_t *n_out) {
  PCLMI_CHECK(n_out != nullptr, PCLMI_ERR_INVALID, "n_out is NULL");
  *n_out = 0;
  PCLMI_CHECK(n == 0 || (in && out), PCLMI_ERR_INVALID, "NULL correspondence array");
  PCLMI_CHECK(in != out || n == 0, PCLMI_ERR_INVALID, "in == out is not allowed");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 correspondences");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  return PCLMI_OK;
}

// count of the last compaction -> host
int finish_count(RejWs *w, size_t *n_out, hipStream_t s) {
  Small h;
  RC(read_small(w, &h, s));
  *n_out = h.count;
  return PCLMI_OK;
}

// host-pointer wrapper: upload -> fn(d_in, d_out, stream) -> download of *n_out records
template <class Fn>
int via_host(const Corr *in, size_t n, Corr *out, size_t *n_out, Fn fn) {
  RC(dev_checks(in, n, out, n_out));
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  RC(g.w->ensure(B_IN, n * sizeof(Corr)));
  RC(g.w->ensure(B_OUT, n * sizeof(Corr)));
  hipStream_t s = g.w->own;
  PCLMI_HIP(hipMemcpyAsync(g.w->buf[B_IN], in, n * sizeof(Corr), hipMemcpyHostToDevice, s));
  RC(fn((const Corr *)g.w->buf[B_IN], (Corr *)g.w->buf[B_OUT], (void *)s));
  if (*n_out) PCLMI_HIP(hipMemcpyAsync(out, g.w->buf[B_OUT], *n_out * sizeof(Corr), hipMemcpyDeviceToHost, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  return PCLMI_OK;
}

}  // namespace
}  // namespace pclmi

using namespace pclmi;

extern "C" {

// ---- device-resident forms ---------------------------------------------------------------------------------------------
int pclmi_reject_distance_dev(const pclmi_correspondence *d_in, size_t n, float max_distance,
                              pclmi_correspondence *d_out, size_t *n_out, void *stream) {
  RC(dev_checks(d_in, n, d_out, n_out));
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  // setMaximumDistance stores the SQUARE (correspondence_rejection_distance.h:88)
  RC(compact(g.w, SrcDistLess{d_in, max_distance * max_distance}, n, d_out, &((Small *)g.w->buf[B_SMALL])->count, s));
  return finish_count(g.w, n_out, s);
}

int pclmi_reject_surface_normal_dev(const pclmi_correspondence *d_in, size_t n, const float *d_src_normals, size_t n_src,
                                    size_t sn_stride_bytes, const float *d_tgt_normals, size_t n_tgt,
                                    size_t tn_stride_bytes, double threshold, pclmi_correspondence *d_out,
                                    size_t *n_out, void *stream) {
  RC(dev_checks(d_in, n, d_out, n_out));
  PCLMI_CHECK(n == 0 || (d_src_normals && d_tgt_normals), PCLMI_ERR_INVALID, "NULL normals");
  PCLMI_CHECK(sn_stride_bytes >= 12 && sn_stride_bytes % 4 == 0 && tn_stride_bytes >= 12 && tn_stride_bytes % 4 == 0,
              PCLMI_ERR_INVALID, "bad stride");
  PCLMI_CHECK(n_src < ((size_t)1 << 31) && n_tgt < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 normals");
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  RC(compact(g.w,
             SrcNormalDot{d_in, d_src_normals, d_tgt_normals, (int)(sn_stride_bytes / 4), (int)(tn_stride_bytes / 4),
                          (int)n_src, (int)n_tgt, threshold},
             n, d_out, &((Small *)g.w->buf[B_SMALL])->count, s));
  return finish_count(g.w, n_out, s);
}

int pclmi_reject_median_dev(const pclmi_correspondence *d_in, size_t n, double factor, pclmi_correspondence *d_out,
                            size_t *n_out, double *median, void *stream) {
  RC(dev_checks(d_in, n, d_out, n_out));
  if (median) *median = 0.0;
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  Small *sm = (Small *)g.w->buf[B_SMALL];
  const float *ds = nullptr;
  RC(sorted_distances(g.w, d_in, n, &ds, s));
  hipLaunchKernelGGL(kth_kernel, dim3(1), dim3(64), 0, s, ds, (int)(n / 2), factor, sm);
  RC(compact(g.w, SrcDistLessEqD{d_in, &sm->b}, n, d_out, &sm->count, s));
  Small h;
  RC(read_small(g.w, &h, s));
  *n_out = h.count;
  if (median) *median = h.a;
  return PCLMI_OK;
}

int pclmi_reject_one_to_one_dev(const pclmi_correspondence *d_in, size_t n, pclmi_correspondence *d_out,
                                size_t *n_out, void *stream) {
  RC(dev_checks(d_in, n, d_out, n_out));
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  Corr *sorted = nullptr;
  RC(sorted_copy(g.w, d_in, n, ByMatchDistQuery(), &sorted, s));
  // the tile counts share B_TMP with the sort's temporary storage: the sort has been enqueued before them
  RC(compact(g.w, SrcRunHead{sorted}, n, d_out, &((Small *)g.w->buf[B_SMALL])->count, s));
  return finish_count(g.w, n_out, s);
}

int pclmi_reject_trimmed_dev(const pclmi_correspondence *d_in, size_t n, float overlap_ratio,
                             unsigned min_correspondences, pclmi_correspondence *d_out, size_t *n_out, void *stream) {
  RC(dev_checks(d_in, n, d_out, n_out));
  if (n == 0) return PCLMI_OK;
  hipStream_t s = (hipStream_t)stream;
  unsigned keep = (unsigned)(int)std::floor(overlap_ratio * (float)n);  // _trimmed.hpp:49
  keep = std::max(keep, min_correspondences);
  if ((size_t)keep >= n) {  // nothing to trim: the input order is kept (_trimmed.hpp:52)
    PCLMI_HIP(hipMemcpyAsync(d_out, d_in, n * sizeof(Corr), hipMemcpyDeviceToDevice, s));
    PCLMI_HIP(hipStreamSynchronize(s));
    *n_out = n;
    return PCLMI_OK;
  }
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  Corr *sorted = nullptr;
  RC(sorted_copy(g.w, d_in, n, ByDistQuery(), &sorted, s));
  PCLMI_HIP(hipMemcpyAsync(d_out, sorted, (size_t)keep * sizeof(Corr), hipMemcpyDeviceToDevice, s));
  PCLMI_HIP(hipStreamSynchronize(s));
  *n_out = keep;
  return PCLMI_OK;
}

// by_distance == 0: the reference AS SHIPPED - its selection loop reads dists[i], which optimizeInlierRatio has sorted
// in place (:87), so the first #(sorted < trimmed) inputs are kept, in input order (:69-76);
// by_distance != 0: the documented intent - every correspondence is judged by its OWN distance.
static int var_trimmed_dev(const Corr *d_in, size_t n, double min_ratio, double max_ratio, double lambda, Corr *d_out,
                           size_t *n_out, double *trim_factor, double *trimmed_distance, int by_distance,
                           hipStream_t s) {
  RC(dev_checks(d_in, n, d_out, n_out));
  if (trim_factor) *trim_factor = 0.0;
  if (trimmed_distance) *trimmed_distance = 0.0;
  if (n == 0) return PCLMI_OK;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  Small *sm = (Small *)g.w->buf[B_SMALL];
  const float *ds = nullptr;
  RC(sorted_distances(g.w, d_in, n, &ds, s));
  const unsigned nbr = (unsigned)n;
  const int min_el = (int)std::floor(min_ratio * nbr), max_el = (int)std::floor(max_ratio * nbr);
  hipLaunchKernelGGL(frms_kernel, dim3(1), dim3(1024), 0, s, ds, (int)n, min_el, max_el, lambda, sm);
  if (by_distance) RC(compact(g.w, SrcDistLessD{d_in, &sm->b}, n, d_out, &sm->count, s));
  Small h;
  RC(read_small(g.w, &h, s));
  if (trim_factor) *trim_factor = h.a;
  if (trimmed_distance) *trimmed_distance = h.b;
  if (by_distance) {
    *n_out = h.count;
  } else {
    *n_out = h.cnt2;
    if (h.cnt2) PCLMI_HIP(hipMemcpyAsync(d_out, d_in, (size_t)h.cnt2 * sizeof(Corr), hipMemcpyDeviceToDevice, s));
    PCLMI_HIP(hipStreamSynchronize(s));
  }
  return PCLMI_OK;
}

int pclmi_reject_var_trimmed_dev(const pclmi_correspondence *d_in, size_t n, double min_ratio, double max_ratio,
                                 double lambda, int by_distance, pclmi_correspondence *d_out, size_t *n_out,
                                 double *trim_factor, double *trimmed_distance, void *stream) {
  return var_trimmed_dev(d_in, n, min_ratio, max_ratio, lambda, d_out, n_out, trim_factor, trimmed_distance,
                         by_distance, (hipStream_t)stream);
}

// CorrespondenceEstimation::determineCorrespondences with source and output resident in HBM: one batched 1-NN
// launch, then the keep test (d2 <= max_distance^2, non-strict, :108) as an order-preserving compaction.
int pclmi_correspondences_dev(const pclmi_tree *target, const float *d_src, size_t n_src, size_t stride_bytes,
                              const int32_t *d_src_indices, size_t n_indices, double max_distance,
                              pclmi_correspondence *d_out, size_t *n_out, void *stream) {
  PCLMI_CHECK(target && n_out, PCLMI_ERR_INVALID, "NULL argument");
  *n_out = 0;
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad stride %zu", stride_bytes);
  const size_t n = d_src_indices ? n_indices : n_src;
  if (n == 0 || target->n == 0) return PCLMI_OK;
  PCLMI_CHECK(d_src && d_out, PCLMI_ERR_INVALID, "NULL source / output");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 source points");
  PCLMI_HIP(hipSetDevice(target->device));
  hipStream_t s = stream ? (hipStream_t)stream : target->stream;
  RejGuard g;
  if (!g.w) return PCLMI_ERR_HIP;
  RC(g.w->ensure(B_SMALL, sizeof(Small)));
  RC(g.w->ensure(B_A, n * sizeof(int)));
  RC(g.w->ensure(B_B, n * sizeof(float)));
  int *nn = (int *)g.w->buf[B_A];
  float *d2 = (float *)g.w->buf[B_B];
  const float *q = d_src;
  size_t q_stride = stride_bytes;
  if (d_src_indices) {  // selected rows: gather them first (x, y, z, 1)
    RC(g.w->ensure(B_IN, n * sizeof(float4)));
    launch_pack(d_src, stride_bytes / 4, d_src_indices, n, (float4 *)g.w->buf[B_IN], s);
    q = (const float *)g.w->buf[B_IN];
    q_stride = 16;
  }
  RC(pclmi_knn_dev_async(target, q, n, q_stride, 1, nn, d2, s));
  RC(compact(g.w, SrcNearest{nn, d2, d_src_indices, max_distance * max_distance}, n, d_out,
             &((Small *)g.w->buf[B_SMALL])->count, s));
  return finish_count(g.w, n_out, s);
}

// ---- host-pointer forms ---------------------------------------------------------------------------------------------------
int pclmi_reject_distance(const pclmi_correspondence *in, size_t n, float max_distance, pclmi_correspondence *out,
                          size_t *n_out) {
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_distance_dev(di, n, max_distance, dout, n_out, s);
  });
}

int pclmi_reject_surface_normal(const pclmi_correspondence *in, size_t n, const float *src_normals, size_t n_src,
                                size_t sn_stride_bytes, const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes,
                                double threshold, pclmi_correspondence *out, size_t *n_out) {
  PCLMI_CHECK(n_out != nullptr, PCLMI_ERR_INVALID, "n_out is NULL");
  *n_out = 0;
  PCLMI_CHECK(n == 0 || (src_normals && tgt_normals), PCLMI_ERR_INVALID, "NULL normals");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  if (n == 0) return PCLMI_OK;
  DevBuf<unsigned char> dsn, dtn;
  RC(dsn.alloc(n_src * sn_stride_bytes));
  RC(dtn.alloc(n_tgt * tn_stride_bytes));
  if (n_src) PCLMI_HIP(hipMemcpy(dsn.p, src_normals, n_src * sn_stride_bytes, hipMemcpyHostToDevice));
  if (n_tgt) PCLMI_HIP(hipMemcpy(dtn.p, tgt_normals, n_tgt * tn_stride_bytes, hipMemcpyHostToDevice));
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_surface_normal_dev(di, n, (const float *)dsn.p, n_src, sn_stride_bytes, (const float *)dtn.p,
                                           n_tgt, tn_stride_bytes, threshold, dout, n_out, s);
  });
}

int pclmi_reject_median(const pclmi_correspondence *in, size_t n, double factor, pclmi_correspondence *out,
                        size_t *n_out, double *median) {
  if (median) *median = 0.0;
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_median_dev(di, n, factor, dout, n_out, median, s);
  });
}

int pclmi_reject_one_to_one(const pclmi_correspondence *in, size_t n, pclmi_correspondence *out, size_t *n_out) {
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_one_to_one_dev(di, n, dout, n_out, s);
  });
}

int pclmi_reject_trimmed(const pclmi_correspondence *in, size_t n, float overlap_ratio, unsigned min_correspondences,
                         pclmi_correspondence *out, size_t *n_out) {
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_trimmed_dev(di, n, overlap_ratio, min_correspondences, dout, n_out, s);
  });
}

int pclmi_reject_var_trimmed(const pclmi_correspondence *in, size_t n, double min_ratio, double max_ratio,
                             double lambda, int by_distance, pclmi_correspondence *out, size_t *n_out,
                             double *trim_factor, double *trimmed_distance) {
  if (trim_factor) *trim_factor = 0.0;
  if (trimmed_distance) *trimmed_distance = 0.0;
  return via_host(in, n, out, n_out, [&](const Corr *di, Corr *dout, void *s) {
    return pclmi_reject_var_trimmed_dev(di, n, min_ratio, max_ratio, lambda, by_distance, dout, n_out, trim_factor,
                                        trimmed_distance, s);
  });
}

}  // extern "C"
