// api_core.hip - C ABI: library/device management, tree lifetime, kNN entry points.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <new>

#include <hipcub/hipcub.hpp>

#include "pclmi_internal.h"

namespace pclmi {

static thread_local char g_err[512] = "";
static thread_local int g_device = 0;

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
    return PCLMI_ERR_NO_DEVICE;
  if (e == hipErrorOutOfMemory) return PCLMI_ERR_OOM;
  return PCLMI_ERR_HIP;
}

// ---- kernel timing ---------------------------------------------------------------------------
namespace {
struct TimingRec {
  int which;
  hipEvent_t a, b;
};
std::mutex g_tmu;
std::vector<TimingRec *> g_pending;
double g_tms[PCLMI_NTIMERS] = {0};
uint64_t g_tcalls[PCLMI_NTIMERS] = {0};
std::atomic<int> g_timing{0};
}  // namespace

TimingScope::TimingScope(int which, hipStream_t stream) : s(stream) {
  const int mask = g_timing.load(std::memory_order_relaxed);
  if (!mask || (mask != 1 && !(mask & (2 << which)))) return;  // 1: every timer; else bit (2 << timer)
  TimingRec *r = new TimingRec{which, nullptr, nullptr};
  if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess ||
      hipEventRecord(r->a, s) != hipSuccess) {
    if (r->a) (void)hipEventDestroy(r->a);
    if (r->b) (void)hipEventDestroy(r->b);
    delete r;
    return;
  }
  rec = r;
}

TimingScope::~TimingScope() {
  if (!rec) return;
  TimingRec *r = (TimingRec *)rec;
  (void)hipEventRecord(r->b, s);
  std::lock_guard<std::mutex> g(g_tmu);
  g_pending.push_back(r);
}

bool nn1_per_lane() {
  static const bool v = [] {
    const char *e = getenv("PCLMI_NN1");
    return !(e && e[0] == 'u');
  }();
  return v;
}

// ---- result-buffer cache ------------------------------------------------------------------------
// hipMalloc / hipFree cost ~0.1 ms each and hipFree synchronises the device; the variable-length
// results of pclmi_radius_dev are therefore recycled: pclmi_free_dev parks up to kCacheSlots buffers
// per process, the next search takes the smallest parked buffer that fits.
namespace {
struct CacheEntry {
  void *p;
  size_t bytes;
  int device;
};
constexpr int kCacheSlots = 8;
constexpr size_t kCacheMaxBytes = (size_t)4 << 30;
std::mutex g_cmu;
std::vector<CacheEntry> g_parked;   // free buffers
std::vector<CacheEntry> g_handed;   // buffers currently owned by callers (size lookup for put)
}  // namespace

void *dev_cache_get(size_t bytes) {
  if (bytes == 0) bytes = 4;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> g(g_cmu);
    int best = -1;
    for (int i = 0; i < (int)g_parked.size(); ++i)
      if (g_parked[i].device == dev && g_parked[i].bytes >= bytes && g_parked[i].bytes <= 2 * bytes + 4096 &&
          (best < 0 || g_parked[i].bytes < g_parked[best].bytes))
        best = i;
    if (best >= 0) {
      CacheEntry e = g_parked[best];
      g_parked.erase(g_parked.begin() + best);
      g_handed.push_back(e);
      return e.p;
    }
  }
  void *p = nullptr;
  const size_t cap = bytes + bytes / 16;  // a little slack so that similar-sized results reuse it
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess) {
    // make room and retry once
    std::vector<CacheEntry> drop;
    {
      std::lock_guard<std::mutex> g(g_cmu);
      drop.swap(g_parked);
    }
    for (auto &d : drop) (void)hipFree(d.p);
    e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
      (void)hip_fail(e, "result buffer allocation", __FILE__, __LINE__);
      return nullptr;
    }
  }
  std::lock_guard<std::mutex> g(g_cmu);
  // a pointer HIP just handed out cannot be live anywhere else: drop any stale record of it (a caller that
  // released a result buffer with hipFree instead of pclmi_free_dev would otherwise poison the size lookup)
  for (size_t i = g_handed.size(); i-- > 0;)
    if (g_handed[i].p == p) g_handed.erase(g_handed.begin() + i);
  g_handed.push_back(CacheEntry{p, cap, dev});
  return p;
}

void dev_cache_put(void *p) {
  if (!p) return;
  CacheEntry e{p, 0, 0};
  bool known = false;
  void *evict = nullptr;
  {
    std::lock_guard<std::mutex> g(g_cmu);
    for (size_t i = 0; i < g_handed.size(); ++i)
      if (g_handed[i].p == p) {
        e = g_handed[i];
        g_handed.erase(g_handed.begin() + i);
        known = true;
        break;
      }
    if (known && e.bytes <= kCacheMaxBytes) {
      if ((int)g_parked.size() >= kCacheSlots) {  // evict the oldest
        evict = g_parked.front().p;
        g_parked.erase(g_parked.begin());
      }
      g_parked.push_back(e);
      p = nullptr;
    }
  }
  if (evict) (void)hipFree(evict);
  if (p) (void)hipFree(p);  // not ours to cache (or too large)
}

static int fixed_too_small(const char *what, size_t need, size_t have) {
  set_error("the allocation-free workspace is too small for this call: %s needs %zu, capacity %zu", what, need, have);
  return PCLMI_ERR_UNSUPPORTED;
}

int Workspace::ensure_queries(size_t nq, size_t nbins) {
  if (fixed) {
    if (nq > cap_q) return fixed_too_small("query slots", nq, cap_q);
    if (nbins > cap_bins) return fixed_too_small("histogram bins", nbins, cap_bins);
    return PCLMI_OK;
  }
  if (nq > cap_q) {
    size_t cap = nq + nq / 8 + 1024;
    if (keys_in) (void)hipFree(keys_in);
    if (ranks) (void)hipFree(ranks);
    if (vals_out) (void)hipFree(vals_out);
    keys_in = nullptr;
    ranks = nullptr;
    vals_out = nullptr;
    cap_q = 0;
    PCLMI_HIP(hipMalloc(&keys_in, cap * 4));
    PCLMI_HIP(hipMalloc(&ranks, cap * 4));
    PCLMI_HIP(hipMalloc(&vals_out, cap * 4));
    cap_q = cap;
  }
  if (nbins > cap_bins) {
    if (hist) (void)hipFree(hist);
    if (sort_tmp) (void)hipFree(sort_tmp);
    hist = nullptr;
    sort_tmp = nullptr;
    cap_bins = 0;
    PCLMI_HIP(hipMalloc(&hist, nbins * 4));
    size_t tmp = 0;
    PCLMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, hist, hist, (int)nbins, (hipStream_t) nullptr));
    cap_tmp = tmp + 256;
    PCLMI_HIP(hipMalloc(&sort_tmp, cap_tmp));
    cap_bins = nbins;
  }
  return PCLMI_OK;
}

int Workspace::ensure_scratch(size_t bytes) {
  if (bytes <= cap_scratch) return PCLMI_OK;
  if (fixed) return fixed_too_small("scratch bytes", bytes, cap_scratch);
  if (scratch) (void)hipFree(scratch);
  scratch = nullptr;
  cap_scratch = 0;
  size_t cap = bytes + bytes / 8 + 4096;
  PCLMI_HIP(hipMalloc(&scratch, cap));
  cap_scratch = cap;
  return PCLMI_OK;
}

int Workspace::ensure_buf(int which, size_t bytes) {
  if (bytes <= cap_buf[which]) return PCLMI_OK;
  if (fixed) return fixed_too_small("buffer bytes", bytes, cap_buf[which]);
  if (buf[which]) (void)hipFree(buf[which]);
  buf[which] = nullptr;
  cap_buf[which] = 0;
  size_t cap = bytes + bytes / 8 + 4096;
  PCLMI_HIP(hipMalloc(&buf[which], cap));
  cap_buf[which] = cap;
  return PCLMI_OK;
}

void Workspace::release() {
  if (ev) (void)hipEventDestroy(ev);
  ev = nullptr;
  if (ev_order) (void)hipEventDestroy(ev_order);
  ev_order = nullptr;
  if (fixed) {  // nothing here is ours
    *this = Workspace();
    return;
  }
  for (int i = 0; i < 4; ++i)
    if (buf[i]) (void)hipFree(buf[i]);
  if (keys_in) (void)hipFree(keys_in);
  if (ranks) (void)hipFree(ranks);
  if (vals_out) (void)hipFree(vals_out);
  if (hist) (void)hipFree(hist);
  if (sort_tmp) (void)hipFree(sort_tmp);
  if (scratch) (void)hipFree(scratch);
  *this = Workspace();
}

Workspace *acquire_ws(const pclmi_tree *t, hipStream_t s) {
  Workspace *w = nullptr;
  {
    std::lock_guard<std::mutex> g(t->mu);
    // prefer a workspace that is idle or whose queued work is on the same stream (stream order is enough then)
    for (size_t i = t->pool.size(); i-- > 0;)
      if (!t->pool[i]->in_flight || t->pool[i]->last == s) {
        w = t->pool[i];
        t->pool.erase(t->pool.begin() + (long)i);
        break;
      }
    if (!w && !t->pool.empty()) {
      w = t->pool.back();
      t->pool.pop_back();
    }
  }
  if (!w) return new Workspace();
  if (w->in_flight && w->last != s && w->ev) (void)hipStreamWaitEvent(s, w->ev, 0);
  return w;
}

void release_ws(const pclmi_tree *t, Workspace *w) {
  w->in_flight = false;
  std::lock_guard<std::mutex> g(t->mu);
  t->pool.push_back(w);
}

void release_ws_async(const pclmi_tree *t, Workspace *w, hipStream_t s) {
  if (!w->ev && hipEventCreateWithFlags(&w->ev, hipEventDisableTiming) != hipSuccess) w->ev = nullptr;
  if (w->ev && hipEventRecord(w->ev, s) == hipSuccess) {
    w->in_flight = true;
    w->last = s;
  } else {
    (void)hipStreamSynchronize(s);  // no event: fall back to waiting here
    w->in_flight = false;
  }
  std::lock_guard<std::mutex> g(t->mu);
  t->pool.push_back(w);
}

int HostSlot::init() {
  PCLMI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  PCLMI_HIP(hipHostMalloc((void **)&h_pin, kPinBytes, hipHostMallocMapped));
  PCLMI_HIP(hipHostGetDevicePointer((void **)&d_pin, h_pin, 0));
  return PCLMI_OK;
}

int HostSlot::ensure(int which, size_t bytes) {
  if (bytes <= cap[which]) return PCLMI_OK;
  if (d[which]) (void)hipFree(d[which]);
  d[which] = nullptr;
  cap[which] = 0;
  const size_t want = bytes + bytes / 8 + 4096;
  PCLMI_HIP(hipMalloc(&d[which], want));
  cap[which] = want;
  return PCLMI_OK;
}

void HostSlot::release() {
  for (int i = 0; i < 4; ++i)
    if (d[i]) (void)hipFree(d[i]);
  if (h_pin) (void)hipHostFree(h_pin);
  if (s) (void)hipStreamDestroy(s);
  *this = HostSlot();
}

HostSlot *acquire_host_slot(const pclmi_tree *t) {
  {
    std::lock_guard<std::mutex> g(t->mu);
    if (!t->host_pool.empty()) {
      HostSlot *h = t->host_pool.back();
      t->host_pool.pop_back();
      return h;
    }
  }
  HostSlot *h = new HostSlot();
  if (h->init() != PCLMI_OK) {
    h->release();
    delete h;
    return nullptr;
  }
  return h;
}

void release_host_slot(const pclmi_tree *t, HostSlot *h) {
  std::lock_guard<std::mutex> g(t->mu);
  t->host_pool.push_back(h);
}

// argument checks + ordering pass + traversal, enqueued on s with workspace w; k_eff == 0 / nq == 0: nothing to do
int knn_enqueue(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_bytes, int k_eff,
                       int32_t *d_idx, float *d_d2, hipStream_t s) {
  const int32_t *order = nullptr;
  RC(launch_morton_order(t, w, d_q, nq, q_stride_bytes / 4, nullptr, nullptr, s, &order));
  return launch_knn(t, w, d_q, nq, q_stride_bytes / 4, order, k_eff, d_idx, d_d2, s);
}

int knn_check(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k, int32_t *d_idx,
                     float *d_d2, int *k_eff) {
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_INVALID, "tree is NULL");
  PCLMI_CHECK(k > 0, PCLMI_ERR_INVALID, "k must be positive (got %d)", k);
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  *k_eff = (size_t)k > t->n ? (int)t->n : k;  // impl/kdtree_flann.hpp:85-86
  if (nq == 0 || *k_eff == 0) return PCLMI_OK;
  PCLMI_CHECK(d_q && d_idx && d_d2, PCLMI_ERR_INVALID, "NULL query/output pointer");
  return PCLMI_OK;
}

}  // namespace pclmi

using namespace pclmi;

// Certificates for "the nearest neighbour has not changed" (TreeView::cert).  For the point t_j of a bucket slot let
// D1 <= D2 be its distances to the nearest and second nearest OTHER indexed points.  A query p with |p - t_j| < 0.49 D1
// has t_j as its nearest neighbour, strictly: every other t_k is at |p - t_k| >= |t_j - t_k| - |p - t_j| >
// (1 / 0.49 - 1) |p - t_j| = 1.04 |p - t_j| - 8 % in the squared distances against the 1e-6 the fp32 evaluations (here,
// in the step kernel and in the walk they replace) can be off by.  With |p - t_j| < 0.49 D2 the same holds for every
// point but the nearest other one, so the answer is the smaller (d2, index) key of those two.  Duplicates of t_j have
// D1 = 0 (level one never fires; level two then decides between the twins by index, as the walk would); three points
// in one place have D2 = 0 and always walk.
#ifndef PCLMI_CERT_FACTOR
#define PCLMI_CERT_FACTOR 0.2401f  // 0.49^2 (a variant built with 1.0f must fail tests/test_gpu_icp.py's certificate test)
#endif
__global__ void slot_of_row_kernel(const float4 *__restrict__ slot, size_t n_slots, size_t rows, int *__restrict__ inv) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const int idx = __float_as_int(slot[s].w);  // external row; -1 in padded slots
  if (idx >= 0 && (size_t)idx < rows) inv[idx] = (int)s;
}

__global__ void cert_from_nn_kernel(const float4 *__restrict__ slot, size_t n_slots, const int *__restrict__ nn,
                                    const float *__restrict__ d2, const int *__restrict__ inv, size_t rows,
                                    float4 *__restrict__ cert, int levels) {
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const int own = __float_as_int(slot[s].w);
  float4 r = make_float4(0.f, 0.f, __int_as_float(0), 0.f);
  if (own >= 0 && (size_t)own < rows) {
    int i[3];
    float d[3];
    for (int u = 0; u < 3; ++u) {
      i[u] = nn[(size_t)own * 3 + u];
      d[u] = d2[(size_t)own * 3 + u];
    }
    // the point itself is one of the three (at distance 0) unless three or more points share its place
    const int self = i[0] == own ? 0 : i[1] == own ? 1 : i[2] == own ? 2 : -1;
    if (self >= 0) {
      const int a = self == 0 ? 1 : 0, b = self == 2 ? 1 : 2;
      const float D1 = d[a], D2 = d[b];
      const int k1 = i[a];
      if (k1 >= 0 && (size_t)k1 < rows && D2 > 0.f && D2 < 3.0e38f && D1 >= 0.f) {
        r.x = __fmul_rn(PCLMI_CERT_FACTOR, D1);
        r.y = levels >= 2 ? __fmul_rn(PCLMI_CERT_FACTOR, D2) : r.x;
        r.z = __int_as_float(inv[k1]);
      }
    }
  }
  cert[s] = r;
}

namespace pclmi {
int tree_ensure_cert(const pclmi_tree *t) {
  // PCLMI_ICP_CERT=0: no certificates (every point walks), =1: the first level only (development switches)
  static const int levels = [] {
    const char *e = getenv("PCLMI_ICP_CERT");
    return e ? atoi(e) : 2;
  }();
  const bool off = levels <= 0;
  // small trees: an iteration there is launch and memory latency whatever the walk does (977-point shard: 33 -> 30 us),
  // and the k = 3 search with its two waits (~0.15 ms) would be a fifth of a whole bun0 -> bun4 align
  if (off || t->n < 4096 || t->rows == 0 || !t->d_cloud) return PCLMI_OK;
  std::lock_guard<std::mutex> g(t->cert_mu);
  if (t->d_cert.load() || t->cert_tried) return PCLMI_OK;
  t->cert_tried = true;
  PCLMI_HIP(hipSetDevice(t->device));
  const size_t rows = t->rows, n_slots = (size_t)t->M * kLeafSize;
  // one block for the three temporaries (hipMalloc / hipFree cost ~0.1-0.5 ms each and hipFree waits for the device)
  DevBuf<int32_t> tmp;
  RC(tmp.alloc(rows * 7));
  struct { int32_t *p; } idx{tmp.p}, inv{tmp.p + rows * 6};
  struct { float *p; } d2{reinterpret_cast<float *>(tmp.p + rows * 3)};
  float *cert = nullptr;
  PCLMI_HIP(hipMalloc(&cert, n_slots * sizeof(float4)));
  // rows that are not indexed (non-finite, or outside an index view) are searched too; nothing reads their result
  int rc = pclmi_knn_dev(t, t->d_cloud, rows, 16, 3, idx.p, d2.p, nullptr);  // waits for its kernels
  if (!rc) {
    const unsigned blocks = (unsigned)((n_slots + 255) / 256);
    hipError_t e = hipMemsetAsync(inv.p, 0, rows * sizeof(int32_t), t->stream);
    hipLaunchKernelGGL(slot_of_row_kernel, dim3(blocks), dim3(256), 0, t->stream, (const float4 *)t->d_slot, n_slots,
                       rows, inv.p);
    hipLaunchKernelGGL(cert_from_nn_kernel, dim3(blocks), dim3(256), 0, t->stream, (const float4 *)t->d_slot, n_slots,
                       idx.p, d2.p, inv.p, rows, (float4 *)cert, levels);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) rc = hip_fail(e, "nearest-other-point certificates", __FILE__, __LINE__);
  }
  if (rc) {
    (void)hipFree(cert);
    return rc;
  }
  t->d_cert.store(cert, std::memory_order_release);
  return PCLMI_OK;
}
}  // namespace pclmi

extern "C" {

const char *pclmi_version(void) { return "pclmi 0.1 (gfx950)"; }
const char *pclmi_last_error(void) { return g_err; }

int pclmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pclmi_set_device(int device) {
  int n = pclmi_device_count();
  PCLMI_CHECK(n > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible");
  PCLMI_CHECK(device >= 0 && device < n, PCLMI_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  PCLMI_HIP(hipSetDevice(device));
  g_device = device;
  return PCLMI_OK;
}

int pclmi_timing_enable(int on) {
  g_timing.store(on < 0 ? 0 : on);
  return PCLMI_OK;
}

int pclmi_timing_pending(void) {
  std::lock_guard<std::mutex> g(g_tmu);
  return (int)g_pending.size();
}

int pclmi_timing_read(int which, double *total_ms, uint64_t *launches, int reset) {
  PCLMI_CHECK(which >= 0 && which < PCLMI_NTIMERS, PCLMI_ERR_INVALID, "unknown timer %d", which);
  std::lock_guard<std::mutex> g(g_tmu);
  for (TimingRec *r : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r->b) == hipSuccess && hipEventElapsedTime(&ms, r->a, r->b) == hipSuccess) {
      g_tms[r->which] += ms;
      g_tcalls[r->which] += 1;
    }
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
  }
  g_pending.clear();
  if (total_ms) *total_ms = g_tms[which];
  if (launches) *launches = g_tcalls[which];
  if (reset) {
    g_tms[which] = 0.0;
    g_tcalls[which] = 0;
  }
  return PCLMI_OK;
}

// development switch PCLMI_TREE_BUILD=host: build the tree with tree_build.cpp and upload it (the device builder,
// tree_build_dev.hip, produces the same tree; tests compare the two)
static bool tree_build_on_host() {
  const char *e = getenv("PCLMI_TREE_BUILD");  // read per call: tests build both ways in one process
  return e && e[0] == 'h';
}

// development aid (not in pclmi.h): the tree's node boxes [2M][8] and leaf records [M][4 * 16] copied to the host;
// box == NULL only reports M
int pclmi_debug_tree_download(const pclmi_tree *t, int *M, float *box, float *leaf) {
  PCLMI_CHECK(t && M, PCLMI_ERR_INVALID, "NULL argument");
  *M = t->M;
  if (!box || !leaf || t->M == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(t->device));
  PCLMI_HIP(hipMemcpy(box, t->d_box, (size_t)2 * t->M * 8 * sizeof(float), hipMemcpyDeviceToHost));
  PCLMI_HIP(hipMemcpy(leaf, t->d_leaf, (size_t)t->M * 4 * kLeafSize * sizeof(float), hipMemcpyDeviceToHost));
  return PCLMI_OK;
}

int pclmi_tree_create_dev(const float *d_xyz, size_t n, size_t stride_bytes, const int32_t *d_indices, size_t n_indices,
                          void *stream, pclmi_tree **out) {
  PCLMI_CHECK(out != nullptr, PCLMI_ERR_INVALID, "out is NULL");
  *out = nullptr;
  PCLMI_CHECK(n == 0 || d_xyz != nullptr, PCLMI_ERR_INVALID, "xyz is NULL");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID,
              "stride_bytes must be a multiple of 4 and >= 12 (got %zu)", stride_bytes);
  PCLMI_CHECK(n < ((size_t)1 << 29) && n_indices < ((size_t)1 << 29), PCLMI_ERR_UNSUPPORTED,
              "the device builder indexes up to 2^29-1 rows");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE, "no HIP device visible: libpclmi has no CPU fallback");
  PCLMI_HIP(hipSetDevice(g_device));
  pclmi_tree *t = new (std::nothrow) pclmi_tree();
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_OOM, "out of host memory");
  t->device = g_device;
  if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
    const int code = hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    pclmi_tree_destroy(t);
    return code;
  }
  // the caller's stream orders the build after whatever produced the cloud; without one the tree's own stream is used
  // (the cloud must then be complete when the call is made)
  const int rc = build_device_tree(t, d_xyz, n, stride_bytes / 4, d_indices, n_indices,
                                   stream ? (hipStream_t)stream : t->stream);
  if (rc) {
    pclmi_tree_destroy(t);
    return rc;
  }
  *out = t;
  return PCLMI_OK;
}

int pclmi_tree_create(const float *xyz, size_t n, size_t stride_bytes, const int32_t *indices, size_t n_indices,
                      int leaf_max, pclmi_tree **out) {
  (void)leaf_max;  // bucket size is a compile-time constant of the kernels (kLeafSize)
  PCLMI_CHECK(out != nullptr, PCLMI_ERR_INVALID, "out is NULL");
  *out = nullptr;
  PCLMI_CHECK(n == 0 || xyz != nullptr, PCLMI_ERR_INVALID, "xyz is NULL");
  PCLMI_CHECK(stride_bytes >= 12 && stride_bytes % 4 == 0, PCLMI_ERR_INVALID,
              "stride_bytes must be a multiple of 4 and >= 12 (got %zu)", stride_bytes);
  PCLMI_CHECK(n < ((size_t)1 << 31), PCLMI_ERR_UNSUPPORTED, "cloud has more than 2^31-1 rows");
  PCLMI_CHECK(pclmi_device_count() > 0, PCLMI_ERR_NO_DEVICE,
              "no HIP device visible: libpclmi has no CPU fallback");
  PCLMI_HIP(hipSetDevice(g_device));
  if (!tree_build_on_host() && n < ((size_t)1 << 29) && n_indices < ((size_t)1 << 29)) {
    // the cloud crosses PCIe once, as the caller laid it out; everything else happens on the device
    DevBuf<unsigned char> raw;
    DevBuf<int32_t> idx;
    RC(raw.alloc(n * stride_bytes));
    if (n) PCLMI_HIP(hipMemcpy(raw.p, xyz, n * stride_bytes, hipMemcpyHostToDevice));
    if (indices) {
      RC(idx.alloc(n_indices));
      if (n_indices) PCLMI_HIP(hipMemcpy(idx.p, indices, n_indices * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return pclmi_tree_create_dev((const float *)raw.p, n, stride_bytes, indices ? idx.p : nullptr, n_indices, nullptr, out);
  }
  const size_t sf = stride_bytes / 4;
  // convertCloudToArray (impl/kdtree_flann.hpp:180-249): keep finite rows, remember their index
  std::vector<float> pts;
  std::vector<int32_t> ext;
  const size_t cand = indices ? n_indices : n;
  pts.reserve(cand * 3);
  ext.reserve(cand);
  for (size_t a = 0; a < cand; ++a) {
    size_t row = indices ? (size_t)indices[a] : a;
    PCLMI_CHECK(row < n, PCLMI_ERR_INVALID, "indices[%zu] = %zu is outside the cloud (%zu rows)", a, row, n);
    const float *p = xyz + row * sf;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    pts.push_back(p[0]);
    pts.push_back(p[1]);
    pts.push_back(p[2]);
    ext.push_back((int32_t)row);
  }
  pclmi_tree *t = new (std::nothrow) pclmi_tree();
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_OOM, "out of host memory");
  t->device = g_device;
  t->n = ext.size();
  t->rows = n;
  int rc = PCLMI_OK;
  auto fail = [&](int code) {
    pclmi_tree_destroy(t);
    return code;
  };
  if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess)
    return fail(hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__));
  if (t->n > 0) {
    HostTree ht;
    build_host_tree(pts.data(), ext.data(), t->n, &ht);
    t->M = ht.M;
    for (int d = 0; d < 3; ++d) {
      t->bb_lo[d] = ht.bb_lo[d];
      t->bb_hi[d] = ht.bb_hi[d];
    }
    {
      double vol = 0.0;
      for (int i = 0; i < ht.M; ++i) {
        const float *b = &ht.box[(size_t)(ht.M + i) * 8];
        vol += (double)(b[3] - b[0]) * (double)(b[4] - b[1]) * (double)(b[5] - b[2]);
      }
      const double bbv = (double)(ht.bb_hi[0] - ht.bb_lo[0]) * (double)(ht.bb_hi[1] - ht.bb_lo[1]) *
                         (double)(ht.bb_hi[2] - ht.bb_lo[2]);
      t->fill = bbv > 0.0 ? (float)(vol / bbv) : 0.f;
    }
    // AoS copy of the buckets for per-lane gathers: slot s = (x, y, z, idx bits)
    std::vector<float> slot((size_t)ht.M * kLeafSize * 4);
    for (size_t sl = 0; sl < (size_t)ht.M * kLeafSize; ++sl) {
      const float *gp = &ht.leaf[(sl / kLeafSize) * 4 * kLeafSize + ((sl % kLeafSize) >> 2) * 16 + (sl & 3)];
      slot[sl * 4 + 0] = gp[0];
      slot[sl * 4 + 1] = gp[4];
      slot[sl * 4 + 2] = gp[8];
      slot[sl * 4 + 3] = gp[12];
    }
    hipError_t e;
    if ((e = hipMalloc(&t->d_slot, slot.size() * 4)) != hipSuccess ||
        (e = hipMemcpy(t->d_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc(&t->d_box, ht.box.size() * 4)) != hipSuccess ||
        (e = hipMalloc(&t->d_leaf, ht.leaf.size() * 4)) != hipSuccess ||
        (e = hipMemcpy(t->d_box, ht.box.data(), ht.box.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(t->d_leaf, ht.leaf.data(), ht.leaf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
      return fail(hip_fail(e, "tree upload", __FILE__, __LINE__));
  }
  if (n > 0) {
    // original rows as (x, y, z, 1): target_ of Registration::setInputTarget (registration.hpp:49-55)
    std::vector<float> cloud(n * 4);
    for (size_t i = 0; i < n; ++i) {
      const float *p = xyz + i * sf;
      cloud[i * 4 + 0] = p[0];
      cloud[i * 4 + 1] = p[1];
      cloud[i * 4 + 2] = p[2];
      cloud[i * 4 + 3] = 1.f;
    }
    hipError_t e;
    if ((e = hipMalloc(&t->d_cloud, n * 16)) != hipSuccess ||
        (e = hipMemcpy(t->d_cloud, cloud.data(), n * 16, hipMemcpyHostToDevice)) != hipSuccess)
      return fail(hip_fail(e, "cloud upload", __FILE__, __LINE__));
  }
  (void)rc;
  *out = t;
  return PCLMI_OK;
}

void pclmi_tree_destroy(pclmi_tree *t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  for (Workspace *w : t->pool) {
    w->release();
    delete w;
  }
  t->pool.clear();
  for (HostSlot *h : t->host_pool) {
    h->release();
    delete h;
  }
  t->host_pool.clear();
  if (t->d_box) (void)hipFree(t->d_box);
  if (t->d_leaf) (void)hipFree(t->d_leaf);
  if (t->d_slot) (void)hipFree(t->d_slot);
  if (t->d_cloud) (void)hipFree(t->d_cloud);
  if (t->d_cert.load()) (void)hipFree(t->d_cert.load());
  if (t->stream) (void)hipStreamDestroy(t->stream);
  if (t->side) (void)hipStreamDestroy(t->side);
  delete t;
}

size_t pclmi_tree_size(const pclmi_tree *t) { return t ? t->n : 0; }


const char *pclmi_knn_kernel_name(const pclmi_tree *t, size_t nq, int k) {
  if (!t || k <= 0) return "";
  return knn_kernel_name(t, nq, (size_t)k > t->n ? (int)t->n : k);
}

int pclmi_knn_dev(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k, int32_t *d_idx,
                  float *d_d2, void *stream) {
  int k_eff = 0;
  RC(knn_check(t, d_q, nq, q_stride_bytes, k, d_idx, d_d2, &k_eff));
  if (nq == 0 || k_eff == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  WsGuard g(t, s);
  RC(knn_enqueue(t, g.w, d_q, nq, q_stride_bytes, k_eff, d_idx, d_d2, s));
  // the workspace goes back to the pool when this returns: wait for the kernels using it
  PCLMI_HIP(hipStreamSynchronize(s));
  return PCLMI_OK;
}

int pclmi_knn_dev_async(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k,
                        int32_t *d_idx, float *d_d2, void *stream) {
  int k_eff = 0;
  RC(knn_check(t, d_q, nq, q_stride_bytes, k, d_idx, d_d2, &k_eff));
  if (nq == 0 || k_eff == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  Workspace *w = acquire_ws(t, s);
  const int rc = knn_enqueue(t, w, d_q, nq, q_stride_bytes, k_eff, d_idx, d_d2, s);
  release_ws_async(t, w, s);  // the next user of w is ordered behind this call's kernels by an event, not the host
  return rc;
}

// The ordering pass (four short, latency-bound kernels: 8 % of a 1M-query k = 16 call) of call i + 1 on the tree's side
// stream, under the traversal of call i; the traversal waits for its ordering pass through an event.  The workspaces of
// consecutive calls alternate (a workspace is reused once the traversal that read it has finished).
int pclmi_knn_dev_pipelined(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k,
                            int32_t *d_idx, float *d_d2, void *stream) {
  int k_eff = 0;
  RC(knn_check(t, d_q, nq, q_stride_bytes, k, d_idx, d_d2, &k_eff));
  if (nq == 0 || k_eff == 0) return PCLMI_OK;
  PCLMI_HIP(hipSetDevice(t->device));
  hipStream_t s = stream ? (hipStream_t)stream : t->stream;
  Workspace *w = nullptr;
  hipStream_t side = nullptr;
  {
    std::lock_guard<std::mutex> g(t->mu);
    if (!t->side) {
      // lowest priority: the pass has a whole traversal's time to finish; its waves should take the slots the
      // traversal leaves, not compete for them
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, lo) != hipSuccess) {
        t->side = nullptr;
        return hip_fail(hipGetLastError(), "hipStreamCreate (side stream)", __FILE__, __LINE__);
      }
    }
    side = t->side;
    for (size_t i = 0; i < t->pool.size(); ++i) {
      Workspace *c = t->pool[i];
      if (!c->in_flight || (c->ev && hipEventQuery(c->ev) == hipSuccess)) {
        w = c;
        w->in_flight = false;
        t->pool.erase(t->pool.begin() + (long)i);
        break;
      }
    }
    if (!w && (t->pool.empty() || t->n_pipe_ws < 3)) {
      w = new Workspace();
      ++t->n_pipe_ws;
    }
    if (!w) {  // every workspace is still being read: the oldest one, behind its traversal
      w = t->pool.front();
      t->pool.erase(t->pool.begin());
    }
  }
  (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
  if (w->in_flight && w->ev) PCLMI_HIP(hipStreamWaitEvent(side, w->ev, 0));
  const int32_t *order = nullptr;
  int rc = launch_morton_order(t, w, d_q, nq, q_stride_bytes / 4, nullptr, nullptr, side, &order);
  if (!rc && order) {
    if (!w->ev_order && hipEventCreateWithFlags(&w->ev_order, hipEventDisableTiming) != hipSuccess) w->ev_order = nullptr;
    if (w->ev_order && hipEventRecord(w->ev_order, side) == hipSuccess) {
      if (hipStreamWaitEvent(s, w->ev_order, 0) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipStreamWaitEvent", __FILE__, __LINE__);
    } else {
      (void)hipStreamSynchronize(side);  // no event: wait here
    }
  }
  if (!rc) rc = launch_knn(t, w, d_q, nq, q_stride_bytes / 4, order, k_eff, d_idx, d_d2, s);
  release_ws_async(t, w, s);
  return rc;
}

int pclmi_knn(const pclmi_tree *t, const float *q, size_t nq, size_t q_stride_bytes, int k, int32_t *idx,
              float *d2, int *k_eff_out) {
  PCLMI_CHECK(t != nullptr, PCLMI_ERR_INVALID, "tree is NULL");
  PCLMI_CHECK(k > 0, PCLMI_ERR_INVALID, "k must be positive (got %d)", k);
  PCLMI_CHECK(q_stride_bytes >= 12 && q_stride_bytes % 4 == 0, PCLMI_ERR_INVALID, "bad query stride %zu",
              q_stride_bytes);
  const int k_eff = (size_t)k > t->n ? (int)t->n : k;
  if (k_eff_out) *k_eff_out = k_eff;
  if (nq == 0 || k_eff == 0) return PCLMI_OK;
  PCLMI_CHECK(q && idx && d2, PCLMI_ERR_INVALID, "NULL query/output pointer");
  PCLMI_HIP(hipSetDevice(t->device));
  HostSlotGuard hg(t);
  if (!hg.h) return PCLMI_ERR_HIP;
  HostSlot *h = hg.h;
  const size_t row = (size_t)k_eff * 4;
  // ---- a few queries (the reference's per-point loop): the kernels read the queries from, and write the results
  //      to, a pinned host buffer in place - one launch and one stream wait, no hipMemcpy, no allocation
  const size_t q_bytes = nq * q_stride_bytes, r_bytes = nq * row;
  const size_t off_i = (q_bytes + 255) & ~(size_t)255, off_d = off_i + ((r_bytes + 255) & ~(size_t)255);
  if (off_d + r_bytes <= HostSlot::kPinBytes) {
    std::memcpy(h->h_pin, q, q_bytes);
    WsGuard g(t, h->s);
    RC(knn_enqueue(t, g.w, (const float *)h->d_pin, nq, q_stride_bytes, k_eff, (int32_t *)(h->d_pin + off_i),
                   (float *)(h->d_pin + off_d), h->s));
    PCLMI_HIP(hipStreamSynchronize(h->s));
    std::memcpy(idx, h->h_pin + off_i, r_bytes);
    std::memcpy(d2, h->h_pin + off_d, r_bytes);
    return PCLMI_OK;
  }
  // ---- batches: grow-only device buffers, chunks of at most ~1 GB of queries + results (4M queries at k = 32:
  //      the batch size of BASELINE configs[4]); the search of chunk c + 1 is queued behind the download of chunk c
  const size_t per_q = q_stride_bytes + 2 * row;
  size_t chunk = ((size_t)1 << 30) / per_q;
  chunk = chunk < 65536 ? 65536 : chunk;
  chunk = chunk > nq ? nq : chunk;
  RC(h->ensure(0, chunk * q_stride_bytes));
  RC(h->ensure(1, chunk * row));
  RC(h->ensure(2, chunk * row));
  WsGuard g(t, h->s);
  for (size_t a = 0; a < nq; a += chunk) {
    const size_t m = nq - a < chunk ? nq - a : chunk;
    PCLMI_HIP(hipMemcpyAsync(h->d[0], (const char *)q + a * q_stride_bytes, m * q_stride_bytes, hipMemcpyHostToDevice,
                             h->s));
    RC(knn_enqueue(t, g.w, (const float *)h->d[0], m, q_stride_bytes, k_eff, (int32_t *)h->d[1], (float *)h->d[2],
                   h->s));
    PCLMI_HIP(hipMemcpyAsync(idx + a * (size_t)k_eff, h->d[1], m * row, hipMemcpyDeviceToHost, h->s));
    PCLMI_HIP(hipMemcpyAsync(d2 + a * (size_t)k_eff, h->d[2], m * row, hipMemcpyDeviceToHost, h->s));
  }
  PCLMI_HIP(hipStreamSynchronize(h->s));
  return PCLMI_OK;
}

void pclmi_free(void *p) { free(p); }
void pclmi_free_dev(void *p) { dev_cache_put(p); }

}  // extern "C"
