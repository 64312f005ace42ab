// api_noalloc.hip - C ABI: the allocation-free k-NN entry point (pclmi_knn_dev_noalloc).
//
// For callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// allocation).  It has no counterpart in the reference.  Same kernels as pclmi_knn_dev; the scratch of the query
// ordering pass (cell ids, ranks, the sorted permutation, the (cell, row) pairs, the tile histograms and the scan's
// temporary storage), which pclmi_knn_dev keeps in hipMalloc'd workspace buffers, lives in ONE file-scope device array.
// Every GPU holds its own copy of the array at its own address; the address is looked up once per device
// (hipGetSymbolAddress, with that device current) and only that pointer is handed to kernels and memsets - the
// array's name is a host-side stand-in, not a device address.
#include <mutex>

#include "pclmi_internal.h"

using namespace pclmi;

namespace {
constexpr size_t kNaMaxQueries = PCLMI_NOALLOC_MAX_QUERIES;
constexpr size_t kNaBins = 160 * 1024;        // two-level order: 512 high bins x 256 tiles + 1; one-level: 4096 cells
constexpr size_t kNaScanTmp = 64 * 1024;      // hipcub::DeviceScan temporary storage for kNaBins counters
constexpr size_t na_align(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t kNaOffKeys = 0;
constexpr size_t kNaOffRanks = kNaOffKeys + na_align(kNaMaxQueries * 4);
constexpr size_t kNaOffVals = kNaOffRanks + na_align(kNaMaxQueries * 4);
constexpr size_t kNaOffPairs = kNaOffVals + na_align(kNaMaxQueries * 4);
constexpr size_t kNaOffHist = kNaOffPairs + na_align(kNaMaxQueries * 8);
constexpr size_t kNaOffTmp = kNaOffHist + na_align(kNaBins * 4);
constexpr size_t kNaBytes = kNaOffTmp + na_align(kNaScanTmp);
constexpr int kNaMaxDevices = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[kNaBytes];  // ~21 MB per GPU, held until the process ends

namespace {
std::mutex g_na_mu;
unsigned char *g_na_ptr[kNaMaxDevices] = {nullptr};
Workspace g_na_ws[kNaMaxDevices];

// device address of this device's copy of `scratch` (looked up on first use, with the device current)
int noalloc_workspace(int device, Workspace **out) {
  PCLMI_CHECK(device >= 0 && device < kNaMaxDevices, PCLMI_ERR_UNSUPPORTED, "device %d: more than %d devices", device,
              kNaMaxDevices);
  std::lock_guard<std::mutex> g(g_na_mu);
  if (!g_na_ptr[device]) {
    PCLMI_HIP(hipSetDevice(device));
    unsigned char *ptr = nullptr;
    PCLMI_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    PCLMI_CHECK(ptr != nullptr, PCLMI_ERR_HIP, "hipGetSymbolAddress returned a null scratch address");
    Workspace &w = g_na_ws[device];
    w.fixed = true;
    w.keys_in = (uint32_t *)(ptr + kNaOffKeys);
    w.ranks = (uint32_t *)(ptr + kNaOffRanks);
    w.vals_out = (int32_t *)(ptr + kNaOffVals);
    w.cap_q = kNaMaxQueries;
    w.scratch = ptr + kNaOffPairs;
    w.cap_scratch = kNaMaxQueries * 8;
    w.hist = (uint32_t *)(ptr + kNaOffHist);
    w.cap_bins = kNaBins;
    w.sort_tmp = ptr + kNaOffTmp;
    w.cap_tmp = kNaScanTmp;
    g_na_ptr[device] = ptr;
  }
  *out = &g_na_ws[device];
  return PCLMI_OK;
}
}  // namespace

extern "C" int pclmi_knn_dev_noalloc(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k,
                          int32_t *d_idx, float *d_d2, void *stream) {
  int k_eff = 0;
  RC(knn_check(t, d_q, nq, q_stride_bytes, k, d_idx, d_d2, &k_eff));
  PCLMI_CHECK(nq <= kNaMaxQueries, PCLMI_ERR_UNSUPPORTED,
              "pclmi_knn_dev_noalloc: %zu queries exceed the static scratch (PCLMI_NOALLOC_MAX_QUERIES = %zu); split "
              "the batch or use pclmi_knn_dev", nq, kNaMaxQueries);
  PCLMI_CHECK(k_eff <= PCLMI_NOALLOC_MAX_K, PCLMI_ERR_UNSUPPORTED,
              "pclmi_knn_dev_noalloc: k = %d exceeds %d (larger k keeps its heaps in allocated scratch: use "
              "pclmi_knn_dev)", k_eff, PCLMI_NOALLOC_MAX_K);
  if (nq == 0 || k_eff == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(t->device));
  Workspace *w = nullptr;
  RC(noalloc_workspace(t->device, &w));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  return knn_enqueue(t, w, d_q, nq, q_stride_bytes, k_eff, d_idx, d_d2, s);
}

