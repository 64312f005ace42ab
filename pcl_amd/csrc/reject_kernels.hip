// reject_kernels.hip - correspondence rejectors on gfx950 (MI355X) + their C ABI.
//
// Replaces the CorrespondenceRejector* family
// (registration/include/pcl/registration/impl/correspondence_rejection_distance.hpp:42-69,
//  _median_distance.hpp:44-69, _one_to_one.hpp:42-67, _trimmed.hpp:42-58, _var_trimmed.hpp:44-107).
// pcl::Correspondence is a 12-byte AoS record; every rejector is a selection over n records:
//   distance / median / var-trimmed : predicate on the distance -> order-preserving compaction
//   one-to-one                      : sort by (match, distance, query), keep the head of each match run
//   trimmed                         : sort by (distance, query), keep the first floor(ratio * n)
// Sorting keys are total orders (ties broken by the query index; the reference's std::sort leaves
// ties unspecified - "parity unpinned" there, defined by the oracle).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "pclmi_internal.h"

namespace pclmi {
namespace {

typedef pclmi_correspondence Corr;

struct DistLess {  // original.distance < limit (correspondence_rejection_distance.hpp:61, strict)
  float limit;
  __host__ __device__ bool operator()(const Corr &c) const { return c.distance < limit; }
};
struct DistLessEqD {  // (double) distance <= limit (correspondence_rejection_median_distance.hpp:64)
  double limit;
  __host__ __device__ bool operator()(const Corr &c) const { return (double)c.distance <= limit; }
};
struct DistLessD {  // (double) distance < limit (correspondence_rejection_var_trimmed.hpp:72)
  double limit;
  __host__ __device__ bool operator()(const Corr &c) const { return (double)c.distance < limit; }
};
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

__global__ void distances_kernel(const Corr *__restrict__ c, int n, float *__restrict__ d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = c[i].distance;
}

// head of each run of equal (non-negative) match indices in a sorted list (_one_to_one.hpp:55-65)
__global__ void run_heads_kernel(const Corr *__restrict__ c, int n, unsigned char *__restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int m = c[i].index_match;
  flag[i] = (m >= 0 && (i == 0 || c[i - 1].index_match != m)) ? 1 : 0;
}

struct Upload {
  DevBuf<Corr> in, out;
  DevBuf<int> num;
  DevBuf<unsigned char> tmp;
  int init(const Corr *h, size_t n) {
    RC(in.alloc(n));
    RC(out.alloc(n));
    RC(num.alloc(1));
    if (n) PCLMI_HIP(hipMemcpy(in.p, h, n * sizeof(Corr), hipMemcpyHostToDevice));
    return PCLMI_OK;
  }
  int download(Corr *h, size_t cnt, const Corr *from) {
    if (cnt) PCLMI_HIP(hipMemcpy(h, from, cnt * sizeof(Corr), hipMemcpyDeviceToHost));
    return PCLMI_OK;
  }
};

template <typename Pred>
int select_if(Upload &u, size_t n, Pred pred, Corr *h_out, size_t *n_out) {
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceSelect::If(nullptr, bytes, u.in.p, u.out.p, u.num.p, (int)n, pred, (hipStream_t) nullptr));
  RC(u.tmp.alloc(bytes + 256));
  bytes = u.tmp.n;
  PCLMI_HIP(hipcub::DeviceSelect::If(u.tmp.p, bytes, u.in.p, u.out.p, u.num.p, (int)n, pred, (hipStream_t) nullptr));
  int cnt = 0;
  PCLMI_HIP(hipMemcpy(&cnt, u.num.p, sizeof(int), hipMemcpyDeviceToHost));
  *n_out = (size_t)cnt;
  return u.download(h_out, (size_t)cnt, u.out.p);
}

template <typename Less>
int sort_records(Upload &u, size_t n, Less less) {
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceMergeSort::SortKeys(nullptr, bytes, u.in.p, (int)n, less, (hipStream_t) nullptr));
  RC(u.tmp.alloc(bytes + 256));
  bytes = u.tmp.n;
  PCLMI_HIP(hipcub::DeviceMergeSort::SortKeys(u.tmp.p, bytes, u.in.p, (int)n, less, (hipStream_t) nullptr));
  return PCLMI_OK;
}

// ascending fp32 distances of the n records, on the host (the O(n log n) part runs on the GPU)
int sorted_distances(Upload &u, size_t n, std::vector<float> &h) {
  DevBuf<float> d, ds;
  RC(d.alloc(n));
  RC(ds.alloc(n));
  hipLaunchKernelGGL(distances_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, u.in.p, (int)n, d.p);
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, d.p, ds.p, (int)n, 0, 32, (hipStream_t) nullptr));
  RC(u.tmp.alloc(bytes + 256));
  bytes = u.tmp.n;
  PCLMI_HIP(hipcub::DeviceRadixSort::SortKeys(u.tmp.p, bytes, d.p, ds.p, (int)n, 0, 32, (hipStream_t) nullptr));
  h.resize(n);
  PCLMI_HIP(hipMemcpy(h.data(), ds.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return PCLMI_OK;
}

int common_checks(const Corr *in, size_t n, Corr *out, size_t *n_out) {
  PCLMI_CHECK(n_out != nullptr, PCLMI_ERR_INVALID, "n_out is NULL");
  *n_out = 0;
  PCLMI_CHECK(n == 0 || (in && out), PCLMI_ERR_INVALID, "NULL correspondence array");
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "more than 2^31-1 correspondences");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  return PCLMI_OK;
}

}  // namespace
}  // namespace pclmi

using namespace pclmi;

extern "C" {

int pclmi_reject_distance(const pclmi_correspondence *in, size_t n, float max_distance, pclmi_correspondence *out,
                          size_t *n_out) {
  RC(common_checks(in, n, out, n_out));
  if (n == 0) return PCLMI_OK;
  Upload u;
  RC(u.init(in, n));
  // setMaximumDistance stores the SQUARE (correspondence_rejection_distance.h:88)
  return select_if(u, n, DistLess{max_distance * max_distance}, out, n_out);
}

int pclmi_reject_median(const pclmi_correspondence *in, size_t n, double factor, pclmi_correspondence *out,
                        size_t *n_out, double *median) {
  RC(common_checks(in, n, out, n_out));
  if (median) *median = 0.0;
  if (n == 0) return PCLMI_OK;
  Upload u;
  RC(u.init(in, n));
  std::vector<float> ds;
  RC(sorted_distances(u, n, ds));
  const double med = (double)ds[n / 2];  // nth_element(n / 2) (_median_distance.hpp:60-61)
  if (median) *median = med;
  return select_if(u, n, DistLessEqD{med * factor}, out, n_out);
}

int pclmi_reject_one_to_one(const pclmi_correspondence *in, size_t n, pclmi_correspondence *out, size_t *n_out) {
  RC(common_checks(in, n, out, n_out));
  if (n == 0) return PCLMI_OK;
  Upload u;
  RC(u.init(in, n));
  RC(sort_records(u, n, ByMatchDistQuery()));
  DevBuf<unsigned char> flag;
  RC(flag.alloc(n));
  hipLaunchKernelGGL(run_heads_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, u.in.p, (int)n,
                     flag.p);
  size_t bytes = 0;
  PCLMI_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, u.in.p, flag.p, u.out.p, u.num.p, (int)n,
                                          (hipStream_t) nullptr));
  RC(u.tmp.alloc(bytes + 256));
  bytes = u.tmp.n;
  PCLMI_HIP(hipcub::DeviceSelect::Flagged(u.tmp.p, bytes, u.in.p, flag.p, u.out.p, u.num.p, (int)n,
                                          (hipStream_t) nullptr));
  int cnt = 0;
  PCLMI_HIP(hipMemcpy(&cnt, u.num.p, sizeof(int), hipMemcpyDeviceToHost));
  *n_out = (size_t)cnt;
  return u.download(out, (size_t)cnt, u.out.p);
}

int pclmi_reject_trimmed(const pclmi_correspondence *in, size_t n, float overlap_ratio, unsigned min_correspondences,
                         pclmi_correspondence *out, size_t *n_out) {
  RC(common_checks(in, n, out, n_out));
  if (n == 0) return PCLMI_OK;
  unsigned keep = (unsigned)(int)std::floor(overlap_ratio * (float)n);  // _trimmed.hpp:49
  keep = std::max(keep, min_correspondences);
  if ((size_t)keep >= n) {  // nothing to trim: the input order is kept (_trimmed.hpp:52)
    std::memcpy(out, in, n * sizeof(pclmi_correspondence));
    *n_out = n;
    return PCLMI_OK;
  }
  Upload u;
  RC(u.init(in, n));
  RC(sort_records(u, n, ByDistQuery()));
  *n_out = keep;
  return u.download(out, keep, u.in.p);
}

int pclmi_reject_var_trimmed(const pclmi_correspondence *in, size_t n, double min_ratio, double max_ratio,
                             double lambda, pclmi_correspondence *out, size_t *n_out, double *trim_factor,
                             double *trimmed_distance) {
  RC(common_checks(in, n, out, n_out));
  if (trim_factor) *trim_factor = 0.0;
  if (trimmed_distance) *trimmed_distance = 0.0;
  if (n == 0) return PCLMI_OK;
  Upload u;
  RC(u.init(in, n));
  std::vector<float> ds;
  RC(sorted_distances(u, n, ds));
  // optimizeInlierRatio (_var_trimmed.hpp:82-107): the O(n) FRMS scan is the host tail of the sort
  const unsigned nbr = (unsigned)n;
  const int min_el = (int)std::floor(min_ratio * nbr), max_el = (int)std::floor(max_ratio * nbr);
  double lower_sum = 0.0;
  for (int i = 0; i < min_el; ++i) lower_sum += (double)ds[i];
  int min_index = 0;
  double best = 0.0;
  for (int i = 0; i < max_el - min_el; ++i) {
    const double id = (double)(min_el + 1 + i);  // LinSpaced(max_el - min_el, min_el + 1, max_el): step 1
    const double ratio = id / (double)nbr;
    const double deno = std::pow(ratio, lambda);
    const double inv = 1.0 / deno;
    const double frms = inv * inv * (1.0 / id) * (lower_sum + (double)ds[min_el + i]);
    if (i == 0 || frms < best) {
      best = frms;
      min_index = i;
    }
  }
  const float opt_ratio = (float)(min_index + min_el) / (float)nbr;
  const double factor = (double)opt_ratio;
  size_t pos = (size_t)(int)((double)n * factor);
  if (pos >= n) pos = n - 1;
  const double trimmed = (double)ds[pos];  // nth_element(int(n * factor)) (:64-65)
  if (trim_factor) *trim_factor = factor;
  if (trimmed_distance) *trimmed_distance = trimmed;
  return select_if(u, n, DistLessD{trimmed}, out, n_out);
}

}  // extern "C"
