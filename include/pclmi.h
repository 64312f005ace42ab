/*
 * pclmi.h - C ABI of the MI355X-native kNN / radius-search / ICP hot path.
 *
 * This is the drop-in boundary: the PCL-shaped C++ facade under include/pcl/ (and the
 * ctypes binding in pcl_amd/) call ONLY these entry points.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference tree).  The reference
 * has no FFI on this path - it is C++ virtual classes over FLANN - so the symbols below
 * are what a binding of that path needs: plain pointers and sizes, int status codes, no
 * C++ types, no exceptions.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *  - Points are float32 xyz at a byte stride (12 for packed xyz, 16 for pcl::PointXYZ,
 *    common/include/pcl/impl/point_types.hpp:141-155).  Only the first three floats are read.
 *  - Entry points without a _dev suffix take HOST pointers and copy; _dev entry points take
 *    DEVICE pointers (inputs already resident in HBM) and enqueue on `stream`
 *    (a hipStream_t passed as void*, NULL = the library's stream for that tree).  The search
 *    entry points (pclmi_knn_dev, pclmi_radius_dev, pclmi_normals_dev) return once their work
 *    on `stream` has completed (their scratch goes back to a pool); pclmi_knn_dev_async and the
 *    ICP step / solve entry points only enqueue.
 *  - Squared L2 distances are fp32, computed as ((dx*dx + dy*dy) + dz*dz) with one rounding
 *    per operation (flann::L2_Simple<float>, kdtree/include/pcl/kdtree/kdtree_flann.h:56).
 *  - Neighbour lists are ascending in the total order (d2, index): exact search, eps = 0,
 *    ties broken by the lowest index.
 *  - Return value: PCLMI_OK (0) or a pclmi_status error; pclmi_last_error() gives text.
 *    There is NO CPU fallback: without a usable HIP device every compute call fails with
 *    PCLMI_ERR_NO_DEVICE.
 *  - All calls taking a `const pclmi_tree*` are thread-safe and re-entrant, as the
 *    reference's const search methods are (features/.../impl/normal_3d_omp.hpp:62-68).
 */
#ifndef PCLMI_H_
#define PCLMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PCLMI_OK = 0,
  PCLMI_ERR_INVALID = 1,     /* bad argument (null pointer, k <= 0, stride < 12, ...) */
  PCLMI_ERR_NO_DEVICE = 2,   /* no HIP device / device init failed */
  PCLMI_ERR_HIP = 3,         /* a HIP runtime call failed */
  PCLMI_ERR_OOM = 4,         /* host or device allocation failed */
  PCLMI_ERR_UNSUPPORTED = 5, /* valid request outside what this build implements */
  PCLMI_ERR_NOT_ENOUGH = 6,  /* ICP: fewer than min_correspondences pairs (icp.hpp:103-108) */
  PCLMI_ERR_COMM = 7         /* multi-GPU exchange: a peer's data did not arrive within the timeout */
} pclmi_status;

typedef struct pclmi_tree pclmi_tree;
typedef struct pclmi_comm pclmi_comm; /* multi-GPU exchange, see below */

/* Library / device ------------------------------------------------------------------ */
const char *pclmi_version(void);
const char *pclmi_last_error(void); /* thread-local text of the last failure */
/* Number of visible HIP devices (0 when none; never fails). */
int pclmi_device_count(void);
/* Select the device used by subsequently created trees in this thread (default 0). */
int pclmi_set_device(int device);

/* Kernel timing -----------------------------------------------------------------------
 * Stands in for the pcl::ScopeTime / StopWatch instrumentation the reference wraps around its
 * searches (common/include/pcl/common/time.h:58-133; test/kdtree/test_kdtree.cpp:137,261).
 * When enabled, the library brackets its dominant kernel of each call (the kNN traversal, the
 * radius count+fill passes, the fused ICP step) with HIP events recorded on the stream the
 * kernel is launched on.  pclmi_timing_read waits for the recorded events, adds their
 * durations to the timer and returns total milliseconds and launch count (reset != 0 zeroes
 * the timer afterwards).  Off by default; costs two event records per bracketed launch when on
 * (a few microseconds each).  pclmi_timing_enable(0) off, (1) every timer, (PCLMI_TIMER_BIT(t) | ...)
 * only the timers named. */
#define PCLMI_TIMER_BIT(t) (2 << (t))
#define PCLMI_TIMER_KNN 0
#define PCLMI_TIMER_RADIUS 1
#define PCLMI_TIMER_ICP_STEP 2
#define PCLMI_TIMER_ORDER 3 /* query ordering (Morton cell counting sort) */
#define PCLMI_NTIMERS 4
int pclmi_timing_enable(int on);
int pclmi_timing_read(int which, double *total_ms, uint64_t *launches, int reset);
/* Recorded event pairs not yet collected by pclmi_timing_read (a watchdog's "is the library waiting for the GPU"). */
int pclmi_timing_pending(void);

/* Index build ------------------------------------------------------------------------
 * Replaces pcl::KdTreeFLANN::setInputCloud + convertCloudToArray
 * (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:46-75,180-249).
 * xyz/n/stride_bytes describe the WHOLE cloud.  indices (nullable, n_indices entries)
 * selects the rows to index, as the IndicesConstPtr overload does (:214-249); returned
 * neighbour indices are always rows of the whole cloud (the reference's index_mapping_).
 * Rows with a non-finite coordinate are skipped (:199-203).  The cloud is copied: later
 * edits do not affect the tree.  leaf_max is accepted for signature compatibility and IGNORED: the bucket size
 * (16 points) is a compile-time constant of the kernels, and no search result depends on it.
 * An empty cloud yields a valid empty tree (searches return 0 neighbours). */
int pclmi_tree_create(const float *xyz, size_t n, size_t stride_bytes, const int32_t *indices,
                      size_t n_indices, int leaf_max, pclmi_tree **out);
/* The same from a cloud that already lives in DEVICE memory (d_xyz / d_indices are device pointers): nothing crosses
 * PCIe.  The reference rebuilds the target tree on every determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation.hpp:47-63), so for a device-resident
 * pipeline the build is on the path.  Built entirely on the GPU (three radix sorts + one stable partition per tree
 * level); pclmi_tree_create uploads the caller's rows once and takes the same route.  The work is enqueued on
 * `stream` (NULL: the tree's own stream - the cloud must then be complete when the call is made); the call returns
 * after the build has finished.  Up to 2^29-1 rows. */
int pclmi_tree_create_dev(const float *d_xyz, size_t n, size_t stride_bytes, const int32_t *d_indices,
                          size_t n_indices, void *stream, pclmi_tree **out);
void pclmi_tree_destroy(pclmi_tree *t);
/* Number of indexed (finite, selected) points = total_nr_points_ (:70). */
size_t pclmi_tree_size(const pclmi_tree *t);

/* k-nearest neighbours ---------------------------------------------------------------
 * Replaces pcl::KdTreeFLANN::nearestKSearch (impl/kdtree_flann.hpp:78-112) and the batched
 * pcl::search::Search::nearestKSearch loops (search/include/pcl/search/search.h:244-262).
 * k is clamped to pclmi_tree_size (:85-86); *k_eff (nullable) receives the clamped value.
 * idx / d2 hold nq * k_eff entries, row-major per query, ascending by (d2, index).
 * Queries must be finite (the reference asserts, :83); a non-finite query yields
 * index -1 / distance NaN rows. */
int pclmi_knn(const pclmi_tree *t, const float *queries, size_t nq, size_t q_stride_bytes, int k,
              int32_t *idx, float *d2, int *k_eff);
/* Same with device-resident queries and outputs.  The row pitch of idx / d2 is
 * min(k, pclmi_tree_size) entries. */
int pclmi_knn_dev(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                  int k, int32_t *d_idx, float *d_d2, void *stream);
/* Same, but only ENQUEUES on `stream` and returns: the batched form of the reference's per-point loop
 * (search.h:244-262) for callers that pipeline uploads, searches and downloads of consecutive batches (the
 * 100M-query sweep of BASELINE configs[4]).  The call's scratch is taken from the tree's workspace pool and handed
 * back in stream order: a later call on another stream is ordered behind this one's kernels with an event, never by
 * blocking the host.  Outputs are valid once the work queued on `stream` has completed. */
int pclmi_knn_dev_async(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                        int k, int32_t *d_idx, float *d_d2, void *stream);
/* pclmi_knn_dev_async for back-to-back batches (the same batched form of search.h:244-262) whose QUERIES ARE ALREADY
 * VALID WHEN THE CALL IS MADE - not produced by work still queued on `stream` (uploaded and synchronised earlier, or
 * resident).  The library may then run the call's query-ordering pass on a stream of its own while the previous
 * call's traversal is still running on `stream`; the traversal itself, and with it the outputs, stay in `stream`
 * order.  Same results as pclmi_knn_dev_async, bit for bit; ~8 % more calls per second at 1M queries, k = 16. */
int pclmi_knn_dev_pipelined(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                            int k, int32_t *d_idx, float *d_d2, void *stream);

/* Name of the traversal kernel a batch of nq queries with this k is dispatched to (for measurement reports). */
const char *pclmi_knn_kernel_name(const pclmi_tree *t, size_t nq, int k);

/* Allocation-free variant of pclmi_knn_dev_async.  It has NO counterpart in the reference: it exists for callers
 * that cannot call hipMalloc - code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same kernels and results; the scratch of the query-ordering pass lives in a file-scope device array of the
 * library (about 21 MB per GPU the library runs on, held until the process ends) instead of pooled hipMalloc'd
 * buffers, so the call performs no hipMalloc / hipFree and creates no events: it only enqueues on `stream`.
 * Limits: nq <= PCLMI_NOALLOC_MAX_QUERIES and min(k, tree size) <= PCLMI_NOALLOC_MAX_K, else
 * PCLMI_ERR_UNSUPPORTED (split the batch, or use pclmi_knn_dev).  Every call on one GPU shares that GPU's array:
 * calls that overlap on one GPU (different streams or threads) RACE - serialise them on one stream.  The first call
 * on a device looks the array's address up (hipGetSymbolAddress); make it once before capturing a graph. */
#define PCLMI_NOALLOC_MAX_QUERIES ((size_t)1 << 20)
#define PCLMI_NOALLOC_MAX_K 32
int pclmi_knn_dev_noalloc(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                          int k, int32_t *d_idx, float *d_d2, void *stream);

/* Radius search ----------------------------------------------------------------------
 * Replaces pcl::KdTreeFLANN::radiusSearch (impl/kdtree_flann.hpp:115-157) and the batched
 * loops (search.h:412-434).  A neighbour qualifies iff d2 < (float)(radius*radius)
 * (strict; the double product cast to float is the reference's, :140).  max_nn == 0 or
 * > size returns all (:125-135), otherwise the nearest max_nn.  sorted != 0: ascending
 * (d2, index) (setSortedResults, kdtree_flann.h:138-144); sorted == 0: ascending index.
 * offsets has nq + 1 entries (CSR).  *idx / *d2 are allocated by the library (host
 * memory, offsets[nq] entries each) and released with pclmi_free. */
int pclmi_radius(const pclmi_tree *t, const float *queries, size_t nq, size_t q_stride_bytes,
                 double radius, uint32_t max_nn, int sorted, uint64_t *offsets, int32_t **idx,
                 float **d2);
/* Device-resident variant: d_offsets has nq + 1 entries; *d_idx / *d_d2 are DEVICE
 * buffers allocated by the library (release with pclmi_free_dev once no queued work still
 * reads them: released buffers are recycled by later searches, not handed back to HIP);
 * *total receives offsets[nq].  Synchronises `stream` (the output size is data dependent). */
int pclmi_radius_dev(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                     double radius, uint32_t max_nn, int sorted, uint64_t *d_offsets,
                     int32_t **d_idx, float **d_d2, uint64_t *total, void *stream);
/* Enqueue-only variant for device-resident pipelines (the same batched radiusSearch, search.h:412-434): the CALLER
 * brings the result arrays - d_idx / d_d2 with room for `capacity` neighbours in all - so nothing waits for the host
 * and consecutive calls queue up (or overlap on two streams).  All neighbours are returned (max_nn = 0).
 * d_status (device, 2 words) receives {offsets[nq], flag}: flag != 0 means the arrays were NOT written - a query
 * has more than 64 neighbours or the total exceeds the capacity - and the search must be repeated with
 * pclmi_radius_dev (d_offsets is valid either way).  Read it once `stream` has reached this point. */
int pclmi_radius_dev_async(const pclmi_tree *t, const float *d_queries, size_t nq, size_t q_stride_bytes,
                           double radius, int sorted, uint64_t *d_offsets, int32_t *d_idx, float *d_d2,
                           uint64_t capacity, uint64_t *d_status, void *stream);
void pclmi_free(void *host_ptr);
void pclmi_free_dev(void *device_ptr);

/* Correspondence estimation ---------------------------------------------------------
 * pcl::Correspondence {index_query, index_match, distance}
 * (common/include/pcl/correspondence.h:53-75). */
typedef struct {
  int32_t index_query;
  int32_t index_match;
  float distance;
} pclmi_correspondence;

/* Replaces CorrespondenceEstimation::determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation.hpp:83-141):
 * 1-NN of every source point in the target tree, kept iff d2 <= max_distance^2
 * (non-strict, :108).  src_indices (nullable) selects source rows (PCLBase indices_).
 * out has capacity n_src (or n_indices); *n_out receives the count, in source order. */
int pclmi_correspondences(const pclmi_tree *target, const float *src, size_t n_src,
                          size_t stride_bytes, const int32_t *src_indices, size_t n_indices,
                          double max_distance, pclmi_correspondence *out, size_t *n_out);
/* Replaces determineReciprocalCorrespondences (:144-233): additionally requires the 1-NN of
 * the matched target point in `source_tree` (built over the same source rows) to be the
 * source point itself.  `tgt` is the target cloud the target tree was built from. */
int pclmi_correspondences_reciprocal(const pclmi_tree *target, const pclmi_tree *source_tree,
                                     const float *src, size_t n_src, size_t src_stride_bytes,
                                     const int32_t *src_indices, size_t n_indices,
                                     const float *tgt, size_t n_tgt, size_t tgt_stride_bytes,
                                     double max_distance, pclmi_correspondence *out,
                                     size_t *n_out);

/* Correspondence rejection -----------------------------------------------------------
 * Replace CorrespondenceRejector*::getRemainingCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_rejection_*.hpp).  `in` and `out` are
 * host arrays (out has capacity n; in == out is not allowed); *n_out receives the remaining count.
 * Where the reference sorts with std::sort (ties unspecified), ties are broken by index_query. */
/* _distance.hpp:42-69: keep distance < max_distance^2 (setMaximumDistance squares,
 * correspondence_rejection_distance.h:88); input order kept. */
int pclmi_reject_distance(const pclmi_correspondence *in, size_t n, float max_distance,
                          pclmi_correspondence *out, size_t *n_out);
/* _median_distance.hpp:44-69: keep distance <= median * factor, median = the n/2-th smallest
 * distance (*median, nullable).  KNOWN DEVIATION: the reference's loop (:62-65) reads the scratch array that
 * std::nth_element has permuted, so WHICH correspondences it keeps depends on the standard library's permutation
 * (the count does not); that cannot be reproduced portably, so every correspondence is judged by its own distance
 * here (the documented intent).  Same count, same median as the reference. */
int pclmi_reject_median(const pclmi_correspondence *in, size_t n, double factor,
                        pclmi_correspondence *out, size_t *n_out, double *median);
/* _one_to_one.hpp:42-67: sort by (index_match, distance), keep the nearest query of every match;
 * output in that sorted order; negative matches are dropped. */
int pclmi_reject_one_to_one(const pclmi_correspondence *in, size_t n, pclmi_correspondence *out,
                            size_t *n_out);
/* CorrespondenceRejectorSurfaceNormal::getRemainingCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_rejection_surface_normal.hpp:44-66 with
 * DataContainer::getCorrespondenceScoreFromNormals, correspondence_rejection.h:254-262): keep a pair iff the fp32 dot
 * product (nx nx' + ny ny') + nz nz' of the source normal at index_query and the target normal at index_match, widened
 * to double, is > threshold (strict; a NaN normal never passes); input order kept.  Normals: three floats at a byte
 * stride (pcl::Normal: 32, pcl::PointNormal: 48 from &normal_x).  A pair whose index lies outside the normal arrays is
 * dropped (the reference reads past its clouds). */
int pclmi_reject_surface_normal(const pclmi_correspondence *in, size_t n, const float *src_normals, size_t n_src,
                                size_t sn_stride_bytes, const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes,
                                double threshold, pclmi_correspondence *out, size_t *n_out);
/* _trimmed.hpp:42-58: keep the max(floor(overlap_ratio * n), min_correspondences) smallest
 * distances, sorted ascending; input returned unchanged when that is >= n. */
int pclmi_reject_trimmed(const pclmi_correspondence *in, size_t n, float overlap_ratio,
                         unsigned min_correspondences, pclmi_correspondence *out, size_t *n_out);
/* _var_trimmed.hpp:44-107: trim factor = first argmin of the FRMS curve over [min_ratio, max_ratio] (defaults 0.05,
 * 0.95, lambda 0.95), trimmed distance = the int(n * factor)-th smallest distance.
 * by_distance == 0: the reference AS SHIPPED - its selection loop (:69-76) reads dists[i], an array that
 * optimizeInlierRatio has sorted in place (:87), so the FIRST #(sorted distances < trimmed) inputs are kept, in input
 * order, whatever their own distances; by_distance != 0: the documented intent - keep the correspondences whose OWN
 * distance is below the trimmed distance, input order kept.  Same count either way.
 * Caveat (parity unpinned): the reference's loop runs over dists[] after std::sort FOLLOWED BY nth_element (:64-78,
 * :87); with several distances EQUAL to the trimmed distance libstdc++'s introselect may move the pivot, so the kept
 * positions need not be a prefix there - which ones the reference keeps then depends on the standard library.  With
 * distinct distances (and always in count) by_distance == 0 is the reference's result. */
int pclmi_reject_var_trimmed(const pclmi_correspondence *in, size_t n, double min_ratio,
                             double max_ratio, double lambda, int by_distance, pclmi_correspondence *out,
                             size_t *n_out, double *trim_factor, double *trimmed_distance);
/* _sample_consensus.hpp:44-110: RANSAC over SampleConsensusModelRegistration (seed 12345) on the
 * pairs (src[index_query], tgt[index_match]); keeps the inliers of the best three-point model in
 * input order, best_T = that model.  Keeps everything (best_T = identity) when no model is found. */
int pclmi_reject_sample_consensus(const float *src, size_t n_src, size_t src_stride_bytes,
                                  const float *tgt, size_t n_tgt, size_t tgt_stride_bytes,
                                  const pclmi_correspondence *in, size_t n, double inlier_threshold,
                                  int max_iterations, pclmi_correspondence *out, size_t *n_out,
                                  float best_T[16]);

/* Device-resident forms of the modular pipeline (test/registration/test_registration_api.cpp:62-347 chain:
 * CorrespondenceEstimation -> CorrespondenceRejector* -> TransformationEstimationSVD): clouds and correspondence
 * records stay in HBM between the stages, work is enqueued on `stream`, scratch comes from a pooled per-device
 * workspace (no hipMalloc in the steady state).  d_in / d_out are DEVICE arrays (d_out capacity n, in != out);
 * only the stage's count (and its scalars) come back: each call waits for its own work on `stream` to read them.
 * Semantics as the host-pointer forms above. */
int pclmi_correspondences_dev(const pclmi_tree *target, const float *d_src, size_t n_src, size_t stride_bytes,
                              const int32_t *d_src_indices, size_t n_indices, double max_distance,
                              pclmi_correspondence *d_out, size_t *n_out, void *stream);
int pclmi_reject_distance_dev(const pclmi_correspondence *d_in, size_t n, float max_distance,
                              pclmi_correspondence *d_out, size_t *n_out, void *stream);
int pclmi_reject_median_dev(const pclmi_correspondence *d_in, size_t n, double factor, pclmi_correspondence *d_out,
                            size_t *n_out, double *median, void *stream);
int pclmi_reject_surface_normal_dev(const pclmi_correspondence *d_in, size_t n, const float *d_src_normals, size_t n_src,
                                    size_t sn_stride_bytes, const float *d_tgt_normals, size_t n_tgt,
                                    size_t tn_stride_bytes, double threshold, pclmi_correspondence *d_out,
                                    size_t *n_out, void *stream);
int pclmi_reject_one_to_one_dev(const pclmi_correspondence *d_in, size_t n, pclmi_correspondence *d_out,
                                size_t *n_out, void *stream);
int pclmi_reject_trimmed_dev(const pclmi_correspondence *d_in, size_t n, float overlap_ratio,
                             unsigned min_correspondences, pclmi_correspondence *d_out, size_t *n_out,
                             void *stream);
int pclmi_reject_var_trimmed_dev(const pclmi_correspondence *d_in, size_t n, double min_ratio, double max_ratio,
                                 double lambda, int by_distance, pclmi_correspondence *d_out, size_t *n_out,
                                 double *trim_factor, double *trimmed_distance, void *stream);
/* TransformationEstimationSVD::estimateRigidTransformation(src, tgt, correspondences, T)
 * (impl/transformation_estimation_svd.hpp:101-111) with clouds and correspondences in HBM: pairs
 * (src[index_query], tgt[index_match]); the 24 fp64 sums are reduced on the GPU, the 3x3 solve runs on the host. */
int pclmi_estimate_rigid_svd_dev(const float *d_src, size_t n_src, size_t src_stride_bytes, const float *d_tgt,
                                 size_t n_tgt, size_t tgt_stride_bytes, const pclmi_correspondence *d_corr, size_t n,
                                 float T[16], void *stream);

/* Rigid transform from paired points ------------------------------------------------
 * Replaces TransformationEstimationSVD::estimateRigidTransformation
 * (registration/include/pcl/registration/impl/transformation_estimation_svd.hpp:81-168):
 * centroids, 3x3 cross-covariance H = sum (s - cs)(t - ct)', SVD, R = V U' with the
 * determinant fix, t = ct - R cs.  The sums are reduced on the GPU in fp64 (one-pass
 * moments), the 3x3 SVD runs on the host in fp64.  T is row-major 4x4 fp32.
 * src_idx / tgt_idx (n entries) index rows of src / tgt (the pcl::Correspondences
 * overload, :101-111); either may be NULL meaning 0..n-1. */
int pclmi_estimate_rigid_svd(const float *src, size_t n_src, size_t src_stride_bytes,
                             const int32_t *src_idx, const float *tgt, size_t n_tgt,
                             size_t tgt_stride_bytes, const int32_t *tgt_idx, size_t n,
                             float T[16]);

/* p <- R p + t in fp32 (common/include/pcl/common/impl/transforms.hpp:168-206); rows with a
 * non-finite coordinate are left untouched (the !is_dense branch, :196-204). */
int pclmi_transform(float *xyz, size_t n, size_t stride_bytes, const float T[16]);

/* Iterative Closest Point ------------------------------------------------------------
 * Replaces pcl::IterativeClosestPoint::align / computeTransformation
 * (registration/include/pcl/registration/impl/registration.hpp:119-168,
 *  registration/include/pcl/registration/impl/icp.hpp:44-211). */
typedef struct {
  int max_iterations;                 /* registration.h:97, default 10 */
  int ransac_iterations;              /* icp.h:104, default 1000; 0 skips the RANSAC step */
  double transformation_epsilon;      /* default 0 */
  double euclidean_fitness_epsilon;   /* default -DBL_MAX (never fires) */
  double max_correspondence_distance; /* default sqrt(DBL_MAX) */
  double ransac_outlier_threshold;    /* inlier_threshold_, default 0.05 */
  int min_correspondences;            /* default 3 */
} pclmi_icp_params;

typedef struct {
  int iterations;
  int converged;
  int last_correspondences; /* pairs under the distance threshold in the last iteration */
  int last_inliers;         /* pairs used for the last transform (after RANSAC) */
  double fitness_delta;     /* mean(d_now - d_prev) of the last iteration (registration.hpp:60-68) */
  double transform_delta;   /* sum |T - T_prev| of the last iteration (icp.hpp:194) */
  float last_transform[16]; /* the last iteration's incremental transform (transformation_, icp.hpp:171):
                               Registration::getLastIncrementalTransformation; identity before any iteration */
} pclmi_icp_stats;

void pclmi_icp_default_params(pclmi_icp_params *p);

/* One call = Registration::align(output, guess).  `target` must have been built from the
 * target cloud (Registration::setInputTarget, registration.hpp:41-57).  src is not
 * modified.  guess may be NULL (identity).  aligned (nullable) receives n_src points at
 * 16-byte stride (x, y, z, 1).  Returns PCLMI_ERR_NOT_ENOUGH when an iteration finds fewer
 * than min_correspondences pairs (converged stays 0, as icp.hpp:103-108). */
int pclmi_icp_align(const pclmi_icp_params *p, const pclmi_tree *target, const float *src,
                    size_t n_src, size_t stride_bytes, const float *guess, float final_T[16],
                    float *aligned, pclmi_icp_stats *stats);

/* Registration::getFitnessScore(max_range) (registration.hpp:71-109): mean squared 1-NN
 * distance of the transformed source, over pairs with d2 <= max_range, fp64 accumulate.
 * Returns DBL_MAX in *score when no pair qualifies. */
int pclmi_fitness_score(const pclmi_tree *target, const float *src, size_t n_src,
                        size_t stride_bytes, const float T[16], double max_range, double *score);

/* Sharded ICP (one process per GPU) ---------------------------------------------------
 * The iteration is split so that a host-side driver can insert ONE all-reduce of the
 * PCLMI_ICP_NSUMS fp64 partial sums between the GPU step and the SVD:
 *   ctx = pclmi_icp_begin(target, src_shard)           uploads + Morton-sorts the shard
 *   loop: pclmi_icp_step_dev(ctx, thr2, d_sums)        apply pending T, 1-NN, threshold,
 *                                                      fp64 block reduce -> d_sums (device)
 *         [all-reduce d_sums across ranks - RCCL via torch.distributed]
 *         pclmi_icp_solve(h_sums, T)                   host: centroids, H, SVD -> T
 *         pclmi_icp_set_transform(ctx, T)              T is applied at the next step / end
 *   pclmi_icp_end(ctx, aligned)                         final transform + download + free
 * pclmi_icp_solve needs no GPU and is the same code pclmi_icp_align runs per iteration. */
#define PCLMI_ICP_NSUMS 24
/* sums layout: [0] n_pairs, [1..3] sum s, [4..6] sum t, [7..15] sum s t' (row-major),
 * [16] sum min(d2, thr2) over ALL points, [17] sum (d_now - d_prev) over ALL points,
 * [18] n_points, [19..23] reserved (zero). */
/* The first ICP context (or align) of a tree searches the tree's cloud against itself once (k = 3) and keeps, per
 * indexed point, 16 bytes of "same neighbour as last iteration" certificates: a source point within 0.49 x the
 * distance from its previous neighbour to that neighbour's nearest (second nearest) other point provably still has
 * that neighbour (or that nearest other point) as its nearest one, and the step skips the tree walk for it - the
 * reference's answer (icp.hpp:84-102) by a shorter route, bit-identical state (PCLMI_ICP_CERT=0 switches it off). */
typedef struct pclmi_icp_ctx pclmi_icp_ctx;
int pclmi_icp_begin(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                    const float *guess, pclmi_icp_ctx **out);
int pclmi_icp_step_dev(pclmi_icp_ctx *ctx, double max_correspondence_distance, double *d_sums,
                       void *stream);
int pclmi_icp_step(pclmi_icp_ctx *ctx, double max_correspondence_distance,
                   double h_sums[PCLMI_ICP_NSUMS]);
int pclmi_icp_solve(const double sums[PCLMI_ICP_NSUMS], float T[16]);
int pclmi_icp_set_transform(pclmi_icp_ctx *ctx, const float T[16]);
int pclmi_icp_end(pclmi_icp_ctx *ctx, float *aligned);

/* Device-resident iteration: the same loop with NO host round trip per iteration.  The solve
 * (centroids, H, 3x3 SVD, T, final = T * final, the convergence criteria of icp.hpp:193-196 and
 * the min-correspondence check of icp.hpp:103-108) runs as a one-lane kernel on the stream, the
 * transform stays in device memory for the next step, and once a criterion has fired every later
 * step / solve of the context is a no-op on the device:
 *   ctx = pclmi_icp_begin(...); pclmi_icp_set_criteria(ctx, params)   (optional: default = never stop)
 *   loop: pclmi_icp_step_dev(ctx, thr, d_sums, stream)
 *         [all-reduce d_sums across ranks on the same stream]
 *         pclmi_icp_solve_dev(ctx, d_sums, stream)
 *   pclmi_icp_state(ctx, final_T, stats)                  synchronises and reads the state back
 *   pclmi_icp_end(ctx, aligned)
 * pclmi_icp_run_dev enqueues `iterations` iterations, each ONE kernel (step, reduction, exchange with the peers
 * when a communicator is attached, solve).  A small shard (up to 64 workgroups = 16k points; PCLMI_ICP_LOOP_MAX=<n>
 * workgroups moves the limit, up to what the chip holds at once, ~262k points) runs them inside ONE launch - the
 * workgroups wait for the next transform instead of ending; same arithmetic in the same order, bit-identical state.  Such launches of one process reserve their workgroups on the
 * device (a context that does not get them runs one launch per iteration); if other work keeps a launch's
 * workgroups off the chip for more than 5 s, pclmi_icp_state returns PCLMI_ERR_HIP (PCLMI_ICP_LOOP=0 disables
 * the resident form).  The device
 * solve is the code of pclmi_icp_solve compiled for the GPU (IEEE fp64, no contraction): same
 * sums, same bits.  A context is driven EITHER this way or with pclmi_icp_set_transform.
 * pclmi_icp_state returns PCLMI_ERR_NOT_ENOUGH when an iteration found fewer pairs than
 * min_correspondences (final_T / stats are still filled). */
int pclmi_icp_set_criteria(pclmi_icp_ctx *ctx, const pclmi_icp_params *p);
int pclmi_icp_solve_dev(pclmi_icp_ctx *ctx, const double *d_sums, void *stream);
int pclmi_icp_run_dev(pclmi_icp_ctx *ctx, double max_correspondence_distance, int iterations,
                      void *stream);
int pclmi_icp_state(pclmi_icp_ctx *ctx, float final_T[16], pclmi_icp_stats *stats);
/* Sharded device-resident loop: with a communicator attached, every iteration of pclmi_icp_run_dev is still ONE
 * kernel per rank - the workgroup that finishes the rank's reduction exchanges the PCLMI_ICP_NSUMS sums with the
 * peers (pclmi_comm, in rank order) and runs the solve; every rank ends with bit-identical transforms.  All ranks
 * must run the same number of iterations (the convergence criteria see identical sums, so they stop together).
 * comm == NULL detaches.  pclmi_icp_state returns PCLMI_ERR_COMM when an exchange timed out. */
int pclmi_icp_set_comm(pclmi_icp_ctx *ctx, pclmi_comm *comm);
/* pclmi_icp_begin for rank `rank` of `world`: every rank passes the WHOLE source; it is ordered along the Hilbert
 * curve once (the same order on every rank) and cut into `world` contiguous runs, so a rank's shard is spatially
 * compact.  pclmi_icp_shard_size: points of this rank's run.  pclmi_icp_end of such a context writes
 * only the rank's own rows of `aligned` (n_src x 16 bytes, rows of the whole source). */
int pclmi_icp_begin_sharded(const pclmi_tree *target, const float *src, size_t n_src, size_t stride_bytes,
                            const float *guess, int rank, int world, pclmi_icp_ctx **out);
size_t pclmi_icp_shard_size(const pclmi_icp_ctx *ctx);
/* pclmi_icp_align over the ranks of `comm` (one process per GPU; every rank calls it with the same arguments and the
 * whole source): target replicated, source sharded as above, one exchange of the PCLMI_ICP_NSUMS sums per iteration
 * inside the step kernel.  final_T and stats are bit-identical on every rank; `aligned` (nullable, n_src rows)
 * receives this rank's own rows only.  The RANSAC rejection step (ransac_iterations > 0, the reference's default)
 * is sharded as well: every rank replays the same sampling stream (same generator, same global pair count), the rank
 * that owns a sampled row contributes its coordinates, the pair mask (only when some row has no pair), the inlier
 * counts of a batch of trials and the refit sums are exchanged inside the kernels - same trials, same inliers, same
 * transform as the single-process align.  The pair mask travels through the communicator's bulk window
 * (PCLMI_COMM_BULK_MB, default 64: sources up to ~16M rows on 8 ranks; PCLMI_ERR_UNSUPPORTED beyond).
 * A communicator of one rank forwards to pclmi_icp_align. */
int pclmi_icp_align_sharded(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                            size_t stride_bytes, const float *guess, pclmi_comm *comm, float final_T[16],
                            float *aligned, pclmi_icp_stats *stats);

/* Point representations in ICP ---------------------------------------------------------------
 * Registration::setPointRepresentation (registration/include/pcl/registration/registration.h:362-370 ->
 * kdtree/include/pcl/kdtree/kdtree.h:204-218 -> impl/kdtree_flann.hpp:91-92): the reference vectorises the moving
 * source point through the representation for every search - the tree indexes the representation of the target - and
 * hands the POINTS THEMSELVES to the estimator, the transform and the convergence criteria.  For the representation
 * "x, y, z times a constant each" (DefaultPointRepresentation / CustomPointRepresentation with setRescaleValues,
 * point_representation.h:141-168; 0 for a dimension the representation leaves out) the device loop does the same:
 *   scaled_target: the tree built from scale * target point (what KdTreeFLANN::setInputCloud builds then);
 *   tgt:           the target cloud itself (rows as the tree's cloud), read by the estimator;
 * distances and the max_correspondence_distance test are in the scaled space (they come from the tree, icp.hpp:93),
 * the moments, the RANSAC step and the transform in the points' own.  SVD estimator only.
 * pclmi_icp_set_search_scale switches a context (before its first step). */
int pclmi_icp_set_search_scale(pclmi_icp_ctx *ctx, const float scale[3], const float *tgt, size_t n_tgt,
                               size_t tgt_stride_bytes);
int pclmi_icp_align_scaled(const pclmi_icp_params *p, const pclmi_tree *scaled_target, const float scale[3],
                           const float *tgt, size_t n_tgt, size_t tgt_stride_bytes, const float *src, size_t n_src,
                           size_t stride_bytes, const float *guess, pclmi_comm *comm /* nullable */, float final_T[16],
                           float *aligned, pclmi_icp_stats *stats);

/* Point-to-plane ICP ------------------------------------------------------------------------
 * IterativeClosestPoint with setTransformationEstimation (TransformationEstimationPointToPlaneLLS)
 * (registration/include/pcl/registration/registration.h:140-160, impl/icp.hpp:171 calling
 * impl/transformation_estimation_point_to_plane_lls.hpp:83-101,141-245): the same correspondence loop, but every
 * iteration solves the linearised point-to-plane problem min sum ((R s + t - d) . n)^2 over the pairs instead of the
 * SVD fit.  The step kernel adds each pair's row to the 21 + 6 sums of the 6x6 normal equations (fp64, the fp32
 * products of :186-188 and :219 as the reference writes them), the launch's last workgroup solves them and builds the
 * transform (constructTransformationMatrix, :118-139) - still ONE kernel per iteration.
 *   tgt_normals: one normal per row of the cloud the target tree was built from (n_tgt >= rows of that cloud);
 *                pairs whose target normal is not finite are skipped by the estimator (:158-174) but still count as
 *                correspondences (icp.hpp:93-98).
 *   src_normals: nullable; the reference also skips pairs whose SOURCE normal is not finite (it never uses its
 *                value, and transformPointCloud leaves it alone): pass the source's normals to reproduce that, NULL
 *                when they are all finite or absent.
 * pclmi_icp_set_point_to_plane switches a context (before its first step); its sums are then
 * PCLMI_ICP_NSUMS_PLANE doubles: [0..20] upper triangle of ATA row by row, [21..26] ATb, [27] n_pairs,
 * [28] sum min(d2, thr2), [29] sum (d_now - d_prev), [30] n_points, [31] zero - pclmi_icp_step_dev writes that many,
 * pclmi_icp_step (24 host doubles) refuses such a context.
 * pclmi_icp_align_point_to_plane = pclmi_icp_align (comm == NULL or one rank) / pclmi_icp_align_sharded with this
 * estimator; the RANSAC rejection step is supported on one GPU (the refit of icp.hpp:171 on the inliers). */
#define PCLMI_ICP_NSUMS_PLANE 32
int pclmi_icp_set_point_to_plane(pclmi_icp_ctx *ctx, const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes,
                                 const float *src_normals, size_t n_src, size_t sn_stride_bytes);
int pclmi_icp_align_point_to_plane(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                                   size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                                   const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes, const float *guess,
                                   pclmi_comm *comm, float final_T[16], float *aligned, pclmi_icp_stats *stats);

/* Per-iteration observer ---------------------------------------------------------------------
 * Registration::registerVisualizationCallback (registration/include/pcl/registration/registration.h,
 * update_visualizer_; called at impl/icp.hpp:182-183 after every iteration with the moved source, the pairs the
 * estimator was given, and the target).  on_iteration receives: the iteration number (1-based), the source as it
 * stands after this iteration's transform (n_src rows of x, y, z, 1 in the caller's row order), and the n_pairs
 * (source row, target row) pairs in ascending source order - with the RANSAC step on, its inliers.  The device
 * loop then runs one iteration per launch with a download in between: an observed align costs PCIe time per
 * iteration; results are those of pclmi_icp_align / _point_to_plane (tgt_normals == NULL: the SVD estimator).
 * One GPU only. */
typedef void (*pclmi_icp_iteration_fn)(void *user, int iteration, const float *cloud, const int32_t *src_idx,
                                       const int32_t *tgt_idx, size_t n_pairs);
int pclmi_icp_align_observed(const pclmi_icp_params *p, const pclmi_tree *target, const float *src, size_t n_src,
                             size_t stride_bytes, const float *src_normals, size_t sn_stride_bytes,
                             const float *tgt_normals, size_t n_tgt, size_t tn_stride_bytes, const float *guess,
                             pclmi_icp_iteration_fn on_iteration, void *user, float final_T[16], float *aligned,
                             pclmi_icp_stats *stats);

/* Multi-GPU exchange (one process per GPU on one node) -----------------------------------
 * The reference is single-process (no counterpart); this is the exchange step of the sharded ICP loop
 * (one all-reduce of the PCLMI_ICP_NSUMS fp64 moments per iteration) kept BELOW the C ABI, so that C++
 * hosts of the facade can use several GPUs without a Python / torch.distributed driver.  Every rank owns a mailbox in
 * device memory that every peer maps through HIP IPC (xGMI peer access); an all-reduce is device code that pushes the
 * values into the peers' mailboxes as 8-byte {sequence tag, payload} granules, waits for theirs and adds in rank order
 * (identical bits on every rank).  Every wait is bounded: a missing peer yields PCLMI_ERR_COMM, never a
 * hung GPU.
 *   pclmi_set_device(local_rank);
 *   pclmi_comm_begin(rank, world, &comm, my_handle);       allocate + export the mailbox
 *   [all-gather the PCLMI_COMM_HANDLE_BYTES handles by any means: MPI, torch.distributed, files]
 *   pclmi_comm_connect(comm, all_handles);                 map the peers' mailboxes
 * or pclmi_comm_create_shm(rank, world, name, &comm): the same with the handles exchanged through the POSIX
 * shared-memory object `name` (unique per job; rank 0 creates it).
 * Every rank must issue the same sequence of exchanges (pclmi_comm_allreduce_f64 calls and ICP iterations of
 * contexts the communicator is attached to).
 * After PCLMI_ERR_COMM the communicator is DEAD: a timed-out exchange leaves the ranks' sequence numbers out of step
 * (a healthy rank finishes exchange s and only times out in s + 1), so every later exchange on it fails at once with
 * PCLMI_ERR_COMM instead of waiting for the timeout again; destroy it and create a new one on every rank.
 * The mailbox lives in uncached (else fine-grained) device memory; when neither can be allocated pclmi_comm_begin
 * fails - coarse-grained memory is not a fallback (the owner's polling loads need not see the peers' writes). */
#define PCLMI_COMM_HANDLE_BYTES 128
#define PCLMI_COMM_MAX_RANKS 16
int pclmi_comm_begin(int rank, int world, pclmi_comm **out, unsigned char handle[PCLMI_COMM_HANDLE_BYTES]);
int pclmi_comm_connect(pclmi_comm *c, const unsigned char *all_handles /* world x PCLMI_COMM_HANDLE_BYTES */);
int pclmi_comm_create_shm(int rank, int world, const char *name, pclmi_comm **out);
int pclmi_comm_rank(const pclmi_comm *c);
int pclmi_comm_size(const pclmi_comm *c);
/* How long a rank waits for a peer inside a kernel before giving up (default 10 s). */
int pclmi_comm_set_timeout(pclmi_comm *c, double seconds);
/* In-place sum over the ranks of d_buf[0..count) (device memory, count <= 64), enqueued on `stream` as ONE
 * one-workgroup kernel.  pclmi_comm_status synchronises the device and reports PCLMI_ERR_COMM if any exchange of
 * this communicator timed out. */
int pclmi_comm_allreduce_f64(pclmi_comm *c, double *d_buf, int count, void *stream);
int pclmi_comm_status(pclmi_comm *c);
void pclmi_comm_destroy(pclmi_comm *c);

/* Surface normals ----------------------------------------------------------------------
 * The largest caller of the batched searches.  Replaces
 * pcl::NormalEstimation<PointInT, PointOutT>::computeFeature
 * (features/include/pcl/features/impl/normal_3d.hpp:46-98) together with the per-point search of
 * Feature::compute (features/include/pcl/features/impl/feature.hpp:143-159,188-220): for every
 * query the k nearest (k > 0, setKSearch) or all points within radius (radius > 0,
 * setRadiusSearch) of the surface tree - exactly one of the two -, the least-squares plane through
 * them (computeMeanAndCovarianceMatrix + solvePlaneParameters, centroid.hpp:553-617,
 * feature.hpp:47-86), curvature = |lambda0 / (lambda0 + lambda1 + lambda2)|, normal flipped towards
 * the viewpoint (flipNormalTowardsViewpoint, normal_3d.h:162-181; NULL = origin).
 * normals holds nq x 4 floats (nx, ny, nz, curvature) - the layout of pcl::Normal's data_n[0..2] and
 * curvature.  Rows of non-finite queries or without neighbours are NaN (normal_3d.hpp:62-68).
 * Moments are accumulated in fp64 about the query point and the eigenproblem is solved in fp64:
 * at least as accurate as the reference's fp32 (which loses the covariance of clouds far from the
 * origin).  With fewer than 3 non-collinear neighbours the plane is not unique and the result is
 * one valid solution, not necessarily the reference's. */
int pclmi_normals(const pclmi_tree *surface, const float *queries, size_t nq, size_t q_stride_bytes,
                  int k, double radius, const float viewpoint[3], float *normals);
int pclmi_normals_dev(const pclmi_tree *surface, const float *d_queries, size_t nq,
                      size_t q_stride_bytes, int k, double radius, const float viewpoint[3],
                      float *d_normals, void *stream);
/* pcl::computePointNormal on one index list (features/include/pcl/features/normal_3d.h:58-104):
 * plane = (nx, ny, nz, d), d = -n . centroid; indices NULL = the whole cloud.  The sign of the
 * normal is not defined by the reference's solver either; flip it towards a viewpoint if needed.
 * Host computation (one list is O(n)); needs no GPU. */
int pclmi_point_normal(const float *pts, size_t n_rows, size_t stride_bytes, const int32_t *indices,
                       size_t n_indices, float plane[4], float *curvature);

/* Point-to-plane pieces -------------------------------------------------
 * Replaces TransformationEstimationPointToPlaneLLS::estimateRigidTransformation
 * (registration/include/pcl/registration/impl/transformation_estimation_point_to_plane_lls.hpp:
 * 45-245): linearised point-to-plane least squares over n pairs (src[src_idx[i]], tgt[tgt_idx[i]]
 * with the target normal tgt_normals[tgt_idx[i]]; index lists nullable = 0..n-1): the 6x6 normal
 * equations are reduced on the GPU in fp64, solved on the host, T = constructTransformationMatrix
 * (alpha, beta, gamma, tx, ty, tz) (:118-139).  Pairs with a non-finite coordinate or normal are
 * skipped (:156-172; src_normals is only consulted for that test and may be NULL).  Returns
 * PCLMI_ERR_NOT_ENOUGH (T = identity) when the system is singular. */
int pclmi_estimate_point_to_plane_lls(const float *src, size_t n_src, size_t src_stride_bytes,
                                      const float *src_normals, size_t sn_stride_bytes,
                                      const int32_t *src_idx, const float *tgt, size_t n_tgt,
                                      size_t tgt_stride_bytes, const float *tgt_normals,
                                      size_t tn_stride_bytes, const int32_t *tgt_idx, size_t n,
                                      float T[16]);
/* Replaces CorrespondenceEstimationNormalShooting::determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation_normal_shooting.hpp:60-165):
 * for every source point, among its k nearest target points (one batched search) the one closest to
 * the line through the source point along its normal - |N x V|^2 in fp64 rounded to fp32, first
 * minimum in ascending-distance order; the pair is dropped when that value exceeds max_distance
 * (compared as the reference compares it: not squared); distance = squared point distance of the
 * chosen neighbour.  out has capacity n_src (or n_indices); pairs come in source order. */
int pclmi_correspondences_normal_shooting(const pclmi_tree *target, const float *src, size_t n_src,
                                          size_t stride_bytes, const float *src_normals,
                                          size_t sn_stride_bytes, const int32_t *src_indices,
                                          size_t n_indices, int k, double max_distance,
                                          pclmi_correspondence *out, size_t *n_out);

/* Replaces CorrespondenceEstimationBackProjection::determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation_backprojection.hpp:58-107):
 * for every source point, among its k nearest target points (one batched search) the arg-min of
 * nn_dists[min_index] * (2 - cos^2(source normal, neighbour normal)) evaluated exactly as the reference loop
 * evaluates it - the weight multiplies the distance of the RUNNING arg-min, and the arg-min is carried over
 * from one source point to the next (:71) - reproduced in parallel by a scan over function composition.
 * The pair is dropped when the minimum exceeds max_distance; distance = squared point distance of the chosen
 * neighbour.  tgt_normals: one row per row of the cloud the tree was built on.  k <= 32.  out has capacity
 * n_src (or n_indices); pairs come in source order. */
int pclmi_correspondences_back_projection(const pclmi_tree *target, const float *src, size_t n_src,
                                          size_t stride_bytes, const float *src_normals,
                                          size_t sn_stride_bytes, const float *tgt_normals, size_t n_tgt,
                                          size_t tn_stride_bytes, const int32_t *src_indices,
                                          size_t n_indices, int k, double max_distance,
                                          pclmi_correspondence *out, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* PCLMI_H_ */
