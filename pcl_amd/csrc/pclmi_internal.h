// pclmi_internal.h - shared declarations of libpclmi.so (not part of the public ABI).
#ifndef PCLMI_INTERNAL_H_
#define PCLMI_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <vector>

#include "comm_dev.hpp"
#include "pclmi.h"

namespace pclmi {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PCLMI_HIP(call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) return ::pclmi::hip_fail(e_, #call, __FILE__, __LINE__);       \
  } while (0)

#define PCLMI_CHECK(cond, code, ...)                                                     \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      ::pclmi::set_error(__VA_ARGS__);                                                   \
      return (code);                                                                     \
    }                                                                                    \
  } while (0)

// ---- kernel timing (pclmi_timing_*): HIP events around the dominant kernels, on their stream ----
struct TimingScope {
  void *rec = nullptr;
  hipStream_t s = nullptr;
  TimingScope(int which, hipStream_t stream);  // no-op unless pclmi_timing_enable(1)
  ~TimingScope();
};

// minimal RAII device buffer
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { reset(); }
  void reset() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  int alloc(size_t count) {
    reset();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    n = count;
    return PCLMI_OK;
  }
};

#define RC(call)              \
  do {                        \
    int rc_ = (call);         \
    if (rc_) return rc_;      \
  } while (0)

// ---- device tree ------------------------------------------------------------------------
// Left-balanced bucket kd-tree in implicit (heap) order: node ids 1 .. 2M-1, children of i
// are 2i and 2i+1, ids >= M are leaves (leaf number = id - M).  See DESIGN.md "Data layout".
#ifndef PCLMI_LEAF
#define PCLMI_LEAF 16
#endif
constexpr int kLeafSize = PCLMI_LEAF;  // points per leaf bucket (L); measured at 1M x 1M, k = 16: 8 -> 0.75 ms, 16 -> 0.66 ms
constexpr float kPadCoord = 3.0e38f; // coordinate of unused bucket slots: d2 overflows to +inf

struct TreeView {
  const float *box;  // [2M][8]  lo.xyz, hi.xyz, 0, 0   (tight bounding box of the subtree)
  const float *leaf; // [M][L/4][16]: per group of 4 points x[4], y[4], z[4], idx[4] (int bits;
                     // external row index, -1 in padded slots) - one 64-byte line per group
  const float *cloud; // [rows][4] original rows (x, y, z, 1), addressed by the external index
  const float *slot;  // [M * L][4] the bucket points again as (x, y, z, idx bits), addressed by
                      // bucket slot = leaf number * L + position: the per-lane gathers of the kNN
                      // buffer compaction stay inside the part of the tree the XCD already holds in L2
  int M;             // number of leaves (>= 1 when n > 0)
  int n;             // number of indexed points
  const float *cert; // [M * L][4] nullable (tree_ensure_cert): per bucket slot j, with D1 <= D2 the distances from that
                     // point to its nearest and second nearest OTHER indexed points: (0.2401 D1^2, 0.2401 D2^2, bucket
                     // slot of the nearest other point as int bits, 0).  A query within 0.49 D1 of t_j has t_j as its one
                     // nearest neighbour; within 0.49 D2, t_j or that nearest other point - the ICP step's certificates
                     // for "same neighbour as last iteration", which replace the walk
  int cert_levels;   // 2: both certificates; 1: the first only (launch_icp_step: grids beyond one resident pass)
};

struct Workspace; // per-call scratch (Morton keys, permutations, sort temp), pooled per tree
struct HostSlot;  // per-call staging of the host-pointer entry points, pooled per tree

}  // namespace pclmi

struct pclmi_tree {
  int device = 0;
  size_t n = 0;          // indexed points
  size_t rows = 0;       // rows of the original cloud
  int M = 0;             // leaves
  float *d_box = nullptr;
  float *d_leaf = nullptr;
  float *d_slot = nullptr;   // [M * L][4] (x, y, z, idx bits) in bucket order
  float *d_cloud = nullptr;  // [rows][4] copy of the original cloud (x, y, z, 1) for ICP gathers
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};
  // volume of the buckets' tight boxes over the volume of the bounding box: ~0.5 for a cloud that fills its box,
  // ~0 for a surface scan.  Only steers the choice of the query order (launch_morton_order); never results.
  float fill = 0.f;
  hipStream_t stream = nullptr;
  // workspace pool (const search calls are re-entrant: each takes a workspace)
  mutable std::mutex mu;
  mutable std::vector<pclmi::Workspace *> pool;
  mutable std::vector<pclmi::HostSlot *> host_pool;
  // pclmi_knn_dev_pipelined: the ordering pass of a call runs on this stream, under the previous call's traversal
  mutable double radius_hint = 0.0;  // neighbours per query of the last radius search (sizes the next call's output)
  mutable hipStream_t side = nullptr;
  mutable int n_pipe_ws = 0;  // workspaces created for pipelined calls (a few alternate)
  // nearest-other-point certificates per bucket slot, computed once on the first ICP context of the tree
  // (tree_ensure_cert)
  mutable std::mutex cert_mu;
  mutable std::atomic<float *> d_cert{nullptr};  // published once, read by every launch's view()
  mutable bool cert_tried = false;

  pclmi::TreeView view() const { return pclmi::TreeView{d_box, d_leaf, d_cloud, d_slot, M, (int)n, d_cert.load(std::memory_order_acquire), 2}; }
};

namespace pclmi {

struct Workspace {
  // query-ordering scratch (counting sort by Morton cell)
  uint32_t *keys_in = nullptr;  // cell id per query
  uint32_t *ranks = nullptr;    // position of the query inside its cell
  int32_t *vals_out = nullptr;  // sorted slot -> query row
  uint32_t *hist = nullptr;     // per-cell counts / running offsets
  void *sort_tmp = nullptr;     // hipcub scan temp storage
  size_t cap_q = 0, cap_bins = 0, cap_tmp = 0;
  // generic scratch (large-k heaps, radius counts, ...)
  void *scratch = nullptr;
  size_t cap_scratch = 0;
  // further independent scratch buffers (radius search: counts/offsets, key arrays, sort temp)
  void *buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap_buf[4] = {0, 0, 0, 0};
  // fixed: the buffers are carved out of memory the workspace does not own (the static scratch of
  // pclmi_knn_dev_noalloc): ensure_* never allocate, a request beyond the capacities fails with PCLMI_ERR_UNSUPPORTED
  bool fixed = false;
  // stream-ordered reuse (pclmi_knn_dev_async): `ev` is recorded on `last` when the workspace goes back to the pool
  // with work still queued; the next user on another stream waits for it on its stream, not on the host
  hipEvent_t ev = nullptr;
  hipEvent_t ev_order = nullptr;  // pipelined calls: the ordering pass on the side stream has finished
  hipStream_t last = nullptr;
  bool in_flight = false;
  int ensure_queries(size_t nq, size_t nbins);
  int ensure_scratch(size_t bytes);
  int ensure_buf(int which, size_t bytes);
  void release();
};

// s: the stream the caller is going to use the workspace on (a workspace released with work still in flight on
// another stream is ordered behind that work with hipStreamWaitEvent)
// Staging of the host-pointer entry points (pclmi_knn, pclmi_radius, pclmi_correspondences ...): a stream of its
// own, grow-only device buffers, and a small host buffer that the GPU reads and writes in place (pinned, mapped), so
// that the steady state performs no hipMalloc / hipFree and a call of a few queries - the reference's normal usage is
// one call per point (search/include/pcl/search/search.h:252-253) - no hipMemcpy either.
struct HostSlot {
  hipStream_t s = nullptr;
  void *d[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  unsigned char *h_pin = nullptr;  // host address of the mapped buffer
  unsigned char *d_pin = nullptr;  // its device address
  static constexpr size_t kPinBytes = (size_t)1 << 20;
  int init();
  int ensure(int which, size_t bytes);
  void release();
};
HostSlot *acquire_host_slot(const pclmi_tree *t);  // nullptr (error text set) on failure
void release_host_slot(const pclmi_tree *t, HostSlot *h);
struct HostSlotGuard {
  const pclmi_tree *t;
  HostSlot *h;
  explicit HostSlotGuard(const pclmi_tree *tt) : t(tt), h(acquire_host_slot(tt)) {}
  ~HostSlotGuard() {
    if (h) release_host_slot(t, h);
  }
};

Workspace *acquire_ws(const pclmi_tree *t, hipStream_t s);
void release_ws(const pclmi_tree *t, Workspace *w);                       // the caller has synchronised its stream
void release_ws_async(const pclmi_tree *t, Workspace *w, hipStream_t s);  // work using w is still queued on s
struct WsGuard {
  const pclmi_tree *t;
  Workspace *w;
  WsGuard(const pclmi_tree *tt, hipStream_t s) : t(tt), w(acquire_ws(tt, s)) {}
  ~WsGuard() { release_ws(t, w); }
};

// ---- host tree builder (tree_build.cpp) --------------------------------------------------
struct HostTree {
  int M = 0;
  std::vector<float> box;   // 2M * 8
  std::vector<float> leaf;  // M * 4 * L
  float bb_lo[3], bb_hi[3];
};
// pts: n packed xyz (finite), ext: n external row indices
void build_host_tree(const float *pts, const int32_t *ext, size_t n, HostTree *out);

// ---- device tree builder (tree_build_dev.hip): the same tree, built from a cloud in device memory --------
int build_device_tree(pclmi_tree *t, const float *d_xyz, size_t n_rows, size_t stride_f, const int32_t *d_indices,
                      size_t n_indices, hipStream_t s);

// ---- kernels launchers (knn_kernels.hip) -------------------------------------------------
// order: nullable device permutation (sorted slot -> query row).  bb_lo / bb_hi: box the Morton
// cells are laid over (NULL = the tree's bounding box).  stable: rows of one cell keep their input order
// (reproducible from run to run; the default ranks them with an atomic counter, which is faster).
int launch_morton_order(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq,
                        size_t q_stride_f, const float *bb_lo, const float *bb_hi, hipStream_t s,
                        const int32_t **d_order, bool stable = false);
const char *knn_kernel_name(const pclmi_tree *t, size_t nq, int k_eff);
// Once per tree (first ICP context): the nearest-other-point certificates of every bucket slot (TreeView::cert), from a
// k = 3 search of the tree's own cloud.  A failure leaves the tree without them (the ICP step then walks for every
// point; same results).
int tree_ensure_cert(const pclmi_tree *t);
int launch_knn(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_f,
               const int32_t *d_order, int k_eff, int32_t *d_idx, float *d_d2, hipStream_t s);
// the search fused with the surface-normal plane fit (k-NN neighbourhoods with knn_normals_fused(k_eff))
bool knn_normals_fused(int k_eff);
int launch_knn_normals(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_f,
                       const int32_t *d_order, int k_eff, const float vp[3], float *d_out, hipStream_t s);


// ---- k-NN entry-point plumbing (api_core.hip), shared by pclmi_knn_dev / _async / _noalloc ----
// knn_check: argument checks, *k_eff = min(k, tree size); knn_enqueue: ordering pass + traversal on s with workspace w
int knn_check(const pclmi_tree *t, const float *d_q, size_t nq, size_t q_stride_bytes, int k, int32_t *d_idx,
              float *d_d2, int *k_eff);
int knn_enqueue(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_bytes, int k_eff,
                int32_t *d_idx, float *d_d2, hipStream_t s);

// ---- small cache of result buffers (api_core.hip): pclmi_free_dev parks, radius search reuses ----
void *dev_cache_get(size_t bytes);  // nullptr on allocation failure (error text set)
void dev_cache_put(void *p);       // nullptr is ignored

// ---- radius search (radius_kernels.hip) ---------------------------------------------------
// Whole device-side pipeline of pclmi_radius_dev; *d_idx / *d_d2 are hipMalloc'd here.
int radius_search_dev(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_f,
                      float r2, uint32_t max_nn, int sorted, uint64_t *d_offsets, int32_t **d_idx,
                      float **d_d2, uint64_t *total, hipStream_t s);

// the radius search fused with the surface-normal plane fit: no neighbour list is materialised
int radius_normals_dev(const pclmi_tree *t, Workspace *w, const float *d_q, size_t nq, size_t q_stride_f, float r2,
                       const float vp[3], float *d_out, hipStream_t s);

// ---- ICP / registration kernels (icp_kernels.hip) -----------------------------------------
struct Xf {
  float m[12];  // rows 0..2 of a row-major 4x4
};
inline Xf make_xf(const float T[16]) {
  Xf x;
  for (int i = 0; i < 12; ++i) x.m[i] = T[i];
  return x;
}
// Device-resident state of one ICP context: lets the iteration run without a host round trip
// (pclmi_icp_solve_dev).  Written by icp_solve_kernel, read by icp_step_kernel.
struct IcpDevState {
  float T[16];        // transform of the last solve; applied by the next step (or at the end) while `apply`
  float final_T[16];  // accumulated transform (icp.hpp:177), starts as the guess
  int apply;          // T waits to be applied
  int done;           // converged or failed: later steps and solves are no-ops
  int status;         // PCLMI_OK / PCLMI_ERR_NOT_ENOUGH
  int iterations;
  int converged;
  int last_pairs;
  int last_inliers;   // pairs the estimator was given (after the RANSAC step; == last_pairs without it)
  int hold;           // RANSAC loop: an iteration waits for the host to run further batches of trials; every
                      // speculatively enqueued kernel of later iterations is a no-op meanwhile
  int max_iterations, min_correspondences;
  double fit, tdelta;
  double transformation_epsilon, euclidean_fitness_epsilon;
  unsigned int ticket;  // the resident loop's iteration counter since the context began: the tag of its hand-offs
                        // (the arrival tickets of the per-iteration launches live in IcpTail)
  // wall_clock64() (100 MHz) of the last step launch: [0] workgroup 0 started, [1] last workgroup arrived,
  // [2] sums ready (after the exchange), [3] solve done.  Development aid (pclmi_debug_icp_ticks).
  unsigned long long ticks[4];
  // -DPCLMI_ICP_TICKS builds only: stamps of wave 0 of the middle workgroup along its walk ([0] start, [1] point
  // loaded, [2] start bound known, [3] common root found, [4] nearest found, [5] results stored, [6] sums in LDS)
  unsigned long long walk[8];
};
// What the workgroup that finishes a step's reduction goes on to do (icp_kernels.hip icp_step_tail): group sums,
// the launch's sums, the exchange with the other ranks (comm.world > 1), the solve (solve != 0, needs st).
// Arrival tickets of different groups sit kTicketStride words (4 KB) apart.  Device-scope atomics execute at the memory
// side, one channel per address range: packed into four cache lines, the ~4000 arrivals of a 1M-point step queued up
// behind each other whenever the workgroups finished together (which the certificates make them do): +6 us per step.
constexpr size_t kTicketStride = 1024;
struct IcpTail {
  double *gpart = nullptr;          // [sums][ngroups] group sums
  unsigned int *gticket = nullptr;  // [ngroups * kTicketStride] arrival tickets (one per group), zero between launches
  unsigned int *ticket = nullptr;   // groups finished, zero between launches
  double *sums_out = nullptr;       // [PCLMI_ICP_NSUMS or _PLANE (+ 8 in the RANSAC loop)] the launch's (all-reduced) sums
  IcpDevState *st = nullptr;
  int solve = 0;
  int ngroups = 1;
  // RANSAC loop, refit launch: the sums that are taken over ALL points (sum d, sum (d - d_prev), point count, origin)
  // come from the step's (already exchanged) sums instead of this launch's
  const double *carry = nullptr;
  // RANSAC loop, first batch of an iteration (ransac_first_kernel): the launch's sums also hold the inlier counts of
  // the batch's models; the tail applies the stopping rule and solves only when the rule settles on the model whose
  // inliers the refit sums were taken over
  struct RansacLoopState *rl = nullptr;
  const struct RansacTrial *trials = nullptr;
  int max_trials = 0;
  // the resident loop (icp_step_kernel<.., LOOP>): rows of tagged granules - block sums, group sums, the publication
  unsigned long long *loopbox = nullptr;
  CommView comm;
};
constexpr int kIcpMaxSums = PCLMI_ICP_NSUMS_PLANE + 8;  // the point-to-plane step of the RANSAC loop
CommView comm_view(const pclmi_comm *c);
int comm_device(const pclmi_comm *c);
bool comm_dead(const pclmi_comm *c);        // an exchange timed out: every later one fails at once
void comm_set_dead(pclmi_comm *c);
const char *comm_memory(const pclmi_comm *c);  // flavour of the mailbox allocation
void launch_transform(float4 *d_pts, size_t n, const Xf &T, hipStream_t s);
void launch_transform_pending(float4 *d_pts, size_t n, IcpDevState *st, hipStream_t s);  // applies st->T iff st->apply
void launch_icp_solve(const double *d_sums, IcpDevState *st, hipStream_t s, bool plane = false);
void launch_pack(const float *d_in, size_t stride_f, const int *d_rows, size_t n, float4 *d_out, hipStream_t s);
void launch_permute(const float4 *in, const int *order, size_t n, float4 *out, hipStream_t s);
void launch_unpermute(const float4 *in, const int *order, size_t n, float4 *out, hipStream_t s);
unsigned icp_step_blocks(size_t n);
unsigned icp_step_groups(size_t n);
// st (nullable): take the pending transform from the device-side state instead of apply / T.  ONE launch: the
// workgroup that arrives last carries out `tail` (sums, exchange, solve).
void launch_icp_step(const pclmi_tree *t, float4 *d_pts, size_t n, bool apply, const Xf &T, double thr2,
                     int *d_nn, int *d_nn_slot, float *d_dist, const double origin[3], double *d_partials,
                     hipStream_t s, IcpDevState *st, bool have_prev, const IcpTail &tail,
                     const float4 *d_tn_slot = nullptr, const unsigned char *d_src_skip = nullptr,
                     unsigned char *d_keep_slot = nullptr /* non-null: the RANSAC loop's step (8 more sums) */,
                     float4 *d_tq = nullptr /* with d_keep_slot: the matched target point per slot */,
                     const float4 *d_raw_tgt = nullptr /* non-null: the tree indexes scale * point (a rescaling point
                                                           representation); the estimator reads these rows */,
                     const float *scale = nullptr,
                     int loop_iters = 1 /* > 1: the resident loop, up to that many iterations in this ONE launch (needs
                                           st, tail.solve, icp_loop_fits(); not with d_keep_slot / d_raw_tgt) */);
size_t icp_loop_capacity(const pclmi_tree *t, bool plane);  // workgroups the chip holds at once (0: no resident loop)
bool icp_loop_fits(const pclmi_tree *t, size_t n, bool plane);
size_t icp_loop_box_words(size_t n);  // 8-byte words of IcpTail::loopbox for a shard of n points
// point-to-plane loop (d_tn_slot != nullptr above): PCLMI_ICP_NSUMS_PLANE sums per launch
void launch_pair_reduce_plane(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, const float4 *tn,
                              const unsigned char *src_skip, const unsigned char *flags, size_t n, double *d_partials,
                              double *d_sums, hipStream_t s);
void launch_normals_to_slots(const pclmi_tree *t, const float4 *tn_rows, float4 *tn_slot, hipStream_t s);
// development switch (environment PCLMI_NN1=union): 1-NN searches use the wave-uniform traversal instead of
// the per-lane one
bool nn1_per_lane();
unsigned pair_reduce_blocks(size_t n);
void launch_pair_reduce(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx,
                        const unsigned char *flags, size_t n, const double origin[3], double *d_partials,
                        double *d_sums, hipStream_t s);
void launch_corr_reduce(const float *src, size_t src_stride_f, const float *tgt, size_t tgt_stride_f,
                        const pclmi_correspondence *corr, size_t n, double *d_partials, double *d_sums, hipStream_t s);
void launch_moment_reduce(const float4 *src, const int *sidx, size_t n, double *d_partials, double *d_sums,
                          hipStream_t s, const unsigned char *keep = nullptr);
void launch_count_within(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, size_t n,
                         const Xf &T, double thr2, unsigned int *d_count, unsigned char *d_flags, hipStream_t s,
                         const unsigned char *keep = nullptr);
void launch_gather_u8(const unsigned char *in, const int *idx, size_t n, unsigned char *out, hipStream_t s);
// ---- RANSAC trials in batches (icp_kernels.hip): the sampling loop of RandomSampleConsensus::computeModel runs on the
// device, up to kRansacBatch trials per launch, and ONE pass over the pairs counts the inliers of all their models
constexpr int kRansacBatch = 64;
struct RansacRng {  // std::mt19937 state (boost::mt19937 in the reference, sac_model.h:95)
  unsigned int mt[624];
  int idx;
};
struct RansacTrial {
  float T[16];
  int status;          // 0: model ready, 1: three-point fit failed (a skipped sample), 2: no good sample in 1000 draws
  unsigned int count;  // inliers of the model (countWithinDistance)
  int pad[2];
};
void launch_ransac_seed(RansacRng *d_rng, unsigned int seed, hipStream_t s);
// draws the next `batch` trials' samples (drawIndexSample + isSampleGood, up to 1000 draws each) and fits their models
// d_sample_dist_thresh: sample_dist_thresh_ of the model, left on the device by launch_ransac_sample_threshold
void launch_ransac_draw(const RansacRng *d_rng_in, RansacRng *d_rng, int *d_shuf, int cnt, const float4 *src,
                        const int *sidx, const float4 *tgt, const int *tidx, const double *d_sample_dist_thresh, int batch,
                        RansacTrial *d_trials, hipStream_t s);
// computeSampleDistanceThreshold (sac_model_registration.h:257-279) from the moment sums of launch_moment_reduce
void launch_ransac_sample_threshold(const double *d_moment_sums, double *d_out, hipStream_t s);
// inliers of every ready model of the batch in one pass over the pairs
void launch_count_within_batch(const float4 *src, const int *sidx, const float4 *tgt, const int *tidx, size_t n,
                               double thr2, int batch, RansacTrial *d_trials, hipStream_t s,
                               const unsigned char *keep = nullptr);
// ---- RANSAC rejection inside the device-resident ICP loop (icp_kernels.hip "RANSAC loop") -------------------------
// Device-side state of the rejection step of ONE iteration (a new, re-seeded model per iteration: icp.hpp:119-120).
struct RansacLoopState {
  unsigned int tag;   // entries of the shuffle array carrying this tag belong to the current iteration's model
  int active;         // this iteration runs the step (>= 3 pairs)
  int cnt;            // pairs of the iteration (the step's sums)
  int all_kept;       // every source row has a pair: position in the pair list == row
  int ntrials;        // models of the current batch
  int bad_run, dead;  // consecutive bad samples of the trial being drawn; 1000 of them: computeModel gives up
  int iterations, best, have, settled;  // ransac.hpp:71-121, carried across the batches of the iteration
  int best_trial;                       // index of the best model inside the batch that produced it
  int need_refit;                       // settled, but not by the model the first batch's fused refit sums belong to
  int batches, trials_last;             // diagnostics: batches of this iteration; trials of the last settled one
  unsigned int skipped, ticket;
  unsigned int wticket, wpad;  // arrivals of the mask-word workgroups of ransac_words_draw_kernel (zero between launches)
  double k, sthr;
  float Tbest[16];
  unsigned long long dbg[12];  // development aid (PCLMI_RANSAC_TICKS=1): wall_clock64 stamps of the last draw kernel
};
struct RansacLoopArgs {
  RansacLoopState *rl;
  IcpDevState *st;
  const double *sums;  // the step's sums (device memory; already exchanged when sharded)
  int est;             // 0: SVD estimator, 1: point-to-plane (layout of sums)
  int n;               // slots of this rank
  int n_rows;          // rows of the whole source: the pair list's length when every point has a pair
  int min_corr, max_trials;
  double inlier_thr2;
  const RansacRng *rng0;
  RansacRng *rng;
  unsigned long long *shuf;  // [n_rows] {tag, value}: shuffled_indices_ without the per-iteration reset
  const int *inv;            // row -> slot (-1: another rank's row)
  const float4 *pts;
  const int *nn_slot;
  const float4 *rows;        // the target's bucket-slot copy (x, y, z, idx bits)
  const float4 *tq;          // per slot: the matched target point (written by the step where keep_slot is set)
  const unsigned char *keep_slot;
  unsigned long long *keep_bits;  // [nsb * 16] pair mask by row, only built when some row has no pair
  unsigned int *sb_cnt, *sb_prefix;  // pairs per 1024 rows, exclusive prefix
  int nsb;
  RansacTrial *trials;
  CommView comm;
};
// first: the first batch of a new iteration (enqueued speculatively behind the step); otherwise a further batch the
// host asked for (st->hold).  draws: samples drawn in this batch (<= kRansacBatch)
void launch_ransac_loop_batch(const RansacLoopArgs &a, int first, int draws, hipStream_t s);
// the first batch's counts fused with the refit sums of its first model, the stopping rule and (usually) the solve
void launch_ransac_loop_first(const RansacLoopArgs &a, const double origin[3], double *d_partials, const IcpTail &tail,
                              const float4 *d_tn_slot, const unsigned char *d_src_skip, hipStream_t s);
// refit on the inliers (or on every pair when no model was found) + the iteration's solve, as the step's tail
void launch_ransac_loop_refit(const RansacLoopArgs &a, const pclmi_tree *t, int first, const double origin[3],
                              double *d_partials, const IcpTail &tail, const float4 *d_tn_slot,
                              const unsigned char *d_src_skip, unsigned char *d_flags, hipStream_t s);

void launch_corr_flags(const int *d_nn, const float4 *d_pts, const float4 *d_cloud, const int *d_order, size_t n,
                       double thr2, int *d_tgt_of_orig, unsigned char *d_keep, hipStream_t s,
                       unsigned char *d_keep_slot = nullptr, const float4 *d_rows = nullptr,
                       const int *d_row_of = nullptr);
void launch_iota(int *v, size_t n, hipStream_t s);
void launch_inverse_permutation(const int *order, size_t n, int *inv, hipStream_t s);  // inv[order[i]] = i
void launch_gather_int(const int *in, const int *idx, size_t n, int *out, hipStream_t s);
void launch_gather_points(const float4 *a, const int *ia, const float4 *b, const int *ib, const int pos[3],
                          float4 *out, hipStream_t s);
int select_flagged(const int *d_in, const unsigned char *d_flags, size_t n, int *d_out, int *d_num, void *d_tmp,
                   size_t tmp_bytes, size_t *need, hipStream_t s);
void launch_fitness(const float *d_d2, size_t n, double max_range, double *d_out2, hipStream_t s);

// ---- surface normals (normal_kernels.hip): idx is CSR (d_offsets != null) or nq rows of k entries ----
void launch_normals(const pclmi_tree *t, const float *d_q, size_t q_stride_f, size_t nq, const int *d_idx,
                    const uint64_t *d_offsets, int k, const float vp[3], float *d_out, hipStream_t s);

// ---- host math (registration_host.cpp; needs no GPU) ----------------------------------------
void svd3_f64(const double A[9], double U[9], double S[3], double V[9]);
int solve_rigid_from_sums(const double sums[PCLMI_ICP_NSUMS], float T[16]);
void mat4_mul_f32(const float A[16], const float B[16], float C[16]);
int solve_point_to_plane(const double sums[27], float T[16]);

}  // namespace pclmi
#endif
