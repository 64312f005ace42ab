/*
 * pcl_oracle.h - CPU restatement of the reference kNN / radius / ICP path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / reported CPU baseline.  The product path (pcl_amd/, include/pclmi.h) never
 * links, imports or falls back to it.
 *
 * Pinning (see oracle/README.md): every function below is checked against the golden
 * vectors the reference's own tests hold for this path (tests/golden/, produced by
 * tests/golden/make_fixtures.py): 397 bun0->bun4 1-NN pairs, 53 reciprocal pairs,
 * 97 distance-rejected pairs, 103 one-to-one pairs, 198 trimmed pairs, 97 RANSAC
 * inliers + transform_from_SAC, transform_from_SVD, the full-ICP 4x4 and the three
 * 10-point kNN known-answer sets.  Tie order ((d2, index) lexicographic) is NOT pinned
 * by any reference test: "parity unpinned" for ties, defined here.
 *
 * Arithmetic: all distances are sequential fp32 ((dx*dx + dy*dy) + dz*dz), one rounding
 * per operation (built with -ffp-contract=off), which is flann::L2_Simple<float>
 * (kdtree/include/pcl/kdtree/kdtree_flann.h:56) without FMA contraction.
 *
 * Points are passed as float arrays with a stride in FLOATS (3 for packed xyz, 4 for
 * PCL's 16-byte PointXYZ, common/include/pcl/impl/point_types.hpp:141-155).
 */
#ifndef PCL_ORACLE_H_
#define PCL_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::Correspondence {index_query, index_match, distance}
 * (common/include/pcl/correspondence.h:53-75). */
typedef struct {
  int32_t index_query;
  int32_t index_match;
  float distance;
} orc_corr;

/* ---- ground-truth searches (search/include/pcl/search/impl/brute_force.hpp:45-136) ---- */

/* Exact kNN by exhaustive scan, results ascending by (d2, index).  k is clamped to n
 * (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:85-86).  Outputs are nq*k_eff,
 * returns k_eff. */
int orc_knn_brute(const float *pts, int64_t n, int pstride, const float *q, int64_t nq,
                  int qstride, int k, int32_t *idx, float *d2);

/* Radius search by exhaustive scan: all points with d2 < r2 (strict, the FLANN rule,
 * impl/kdtree_flann.hpp:137-141); max_nn==0 or >n -> all (:125-135), else the nearest
 * max_nn; ascending by (d2,index) when sorted!=0, ascending index otherwise.
 * offsets has nq+1 entries; *idx / *d2 are malloc'd (free with orc_free). */
int orc_radius_brute(const float *pts, int64_t n, int pstride, const float *q, int64_t nq,
                     int qstride, float r2, uint32_t max_nn, int sorted, int64_t *offsets,
                     int32_t **idx, float **d2);
void orc_free(void *p);

/* ---- FLANN-style single kd-tree (CPU baseline; results identical to brute force) ---- */
typedef struct orc_tree orc_tree;
orc_tree *orc_tree_build(const float *pts, int64_t n, int pstride, int leaf_max);
void orc_tree_free(orc_tree *t);
/* nthreads<=1: serial per-query loop (search/include/pcl/search/search.h:252-253);
 * otherwise OpenMP over queries. */
int orc_tree_knn(const orc_tree *t, const float *q, int64_t nq, int qstride, int k,
                 int32_t *idx, float *d2, int nthreads);
int orc_tree_radius(const orc_tree *t, const float *q, int64_t nq, int qstride, float r2,
                    uint32_t max_nn, int sorted, int64_t *offsets, int32_t **idx, float **d2,
                    int nthreads);

/* ---- registration pieces ---- */

/* p <- R*p + t in fp32, in place (common/include/pcl/common/impl/transforms.hpp:184-191).
 * T is row-major 4x4. */
void orc_transform(float *pts, int64_t n, int stride, const float T[16]);

/* TransformationEstimationSVD on n already-paired points (fp32 two-pass centroid /
 * demean / H = S*T' / Jacobi SVD / det fix), row-major 4x4 out
 * (registration/include/pcl/registration/impl/transformation_estimation_svd.hpp:114-168). */
void orc_estimate_rigid_svd(const float *src, int sstride, const int32_t *sidx,
                            const float *tgt, int tstride, const int32_t *tidx, int64_t n,
                            float T[16]);
/* Same estimator with the reference's Scalar template parameter = double
 * (transformation_estimation_svd.h:57): every accumulation and the SVD in fp64. */
void orc_estimate_rigid_svd_f64(const float *src, int sstride, const int32_t *sidx,
                                const float *tgt, int tstride, const int32_t *tidx, int64_t n,
                                float T[16]);

/* CorrespondenceEstimation::determine[Reciprocal]Correspondences
 * (registration/include/pcl/registration/impl/correspondence_estimation.hpp:83-141,144-233);
 * keeps d2 <= max_dist^2.  out must hold ns entries; returns the count. */
int64_t orc_correspondences(const float *src, int64_t ns, int sstride, const float *tgt,
                            int64_t nt, int tstride, double max_dist, int reciprocal,
                            orc_corr *out);

/* Rejectors (registration/include/pcl/registration/impl/correspondence_rejection_*.hpp).
 * All write to out (capacity n) and return the remaining count. */
int64_t orc_reject_distance(const orc_corr *in, int64_t n, float max_dist, orc_corr *out);
int64_t orc_reject_surface_normal(const orc_corr *in, int64_t n, const float *sn, int sns, const float *tn, int tns,
                                  double threshold, orc_corr *out);
int64_t orc_reject_median(const orc_corr *in, int64_t n, double factor, orc_corr *out,
                          double *median);
int64_t orc_reject_one_to_one(const orc_corr *in, int64_t n, orc_corr *out);
int64_t orc_reject_trimmed(const orc_corr *in, int64_t n, float overlap, unsigned min_corr,
                           orc_corr *out);
int64_t orc_reject_var_trimmed(const orc_corr *in, int64_t n, double min_ratio, double max_ratio,
                               double lambda, int by_distance, orc_corr *out, double *factor,
                               double *trimmed_distance);
/* RANSAC over SampleConsensusModelRegistration, seed 12345
 * (sample_consensus/include/pcl/sample_consensus/impl/ransac.hpp:44-133,
 *  impl/sac_model_registration.hpp, sac_model.h:87-97,166-198,401-412).
 * Returns number of inliers (positions into in[]) or -1 when computeModel() fails. */
int64_t orc_ransac_registration(const float *src, int64_t ns, int sstride, const float *tgt,
                                int64_t nt, int tstride, const int32_t *sidx,
                                const int32_t *tidx, int64_t n, double inlier_threshold,
                                int max_iterations, int32_t *inlier_pos, float T[16],
                                int *trials);

typedef struct {
  int max_iterations;            /* registration.h:97 default 10 */
  int ransac_iterations;         /* icp.h:104 default 1000; 0 disables the RANSAC step */
  double transformation_epsilon; /* default 0 */
  double euclidean_fitness_epsilon; /* default -DBL_MAX */
  double max_correspondence_distance; /* default sqrt(DBL_MAX) */
  double ransac_outlier_threshold;    /* inlier_threshold_, default 0.05 */
  int min_correspondences;       /* default 3 */
  int svd_scalar_double;         /* 0: TransformationEstimationSVD<.., float> (default); 1: <.., double> */
} orc_icp_params;

typedef struct {
  int iterations;
  int converged;
  int last_correspondences;
  double fitness_delta;
} orc_icp_stats;

void orc_icp_default_params(orc_icp_params *p);

/* IterativeClosestPoint::align (registration/include/pcl/registration/impl/registration.hpp:119-168,
 * impl/icp.hpp:44-211).  src is not modified; aligned (ns points, stride 4, w=1) receives the
 * registered cloud; final_T row-major.  guess may be NULL (identity). */
int orc_icp_align(const orc_icp_params *p, const float *src, int64_t ns, int sstride,
                  const float *tgt, int64_t nt, int tstride, const float *guess,
                  float final_T[16], float *aligned, orc_icp_stats *stats, int nthreads);

/* The same loop with Registration::setPointRepresentation (registration.h:362-370) for a representation that
 * rescales x, y, z (point_representation.h:141-168): the tree indexes scale * target point, every search vectorises
 * the moved source point the same way (kdtree.h:204-218, impl/kdtree_flann.hpp:91-92); the estimator, RANSAC and
 * the transform read the points themselves. */
int orc_icp_align_scaled(const orc_icp_params *p, const float scale[3], const float *src, int64_t ns, int sstride,
                         const float *tgt, int64_t nt, int tstride, const float *guess, float final_T[16],
                         float *aligned, orc_icp_stats *stats, int nthreads);

/* The same loop with setTransformationEstimation (TransformationEstimationPointToPlaneLLS)
 * (registration/include/pcl/registration/registration.h:140-160; icp.hpp:171 then calls
 * impl/transformation_estimation_point_to_plane_lls.hpp:83-101,141-245 on the *_good index lists).
 * tn: target normals (row i of tn belongs to row i of tgt); sn: source normals, nullable (only tested for
 * finiteness, :158-174). */
int orc_icp_align_point_to_plane(const orc_icp_params *p, const float *src, int64_t ns, int sstride,
                                 const float *sn, int snstride, const float *tgt, int64_t nt, int tstride,
                                 const float *tn, int tnstride, const float *guess, float final_T[16],
                                 float *aligned, orc_icp_stats *stats, int nthreads);

/* Registration::getFitnessScore(max_range) (impl/registration.hpp:71-109). */
double orc_fitness_score(const float *src, int64_t ns, int sstride, const float *tgt,
                         int64_t nt, int tstride, const float final_T[16], double max_range);

/* ---- surface normals (the largest caller of the batched searches) ---- */

/* computePointNormal on an index list (features/include/pcl/features/normal_3d.h:88-104):
 * computeMeanAndCovarianceMatrix in fp32, sequential in list order
 * (common/include/pcl/common/impl/centroid.hpp:553-617), then solvePlaneParameters
 * (features/include/pcl/features/impl/feature.hpp:47-86) = pcl::eigen33 closed-form smallest
 * eigenpair (common/include/pcl/common/eigen.h:105-190,334-368) and curvature
 * |lambda0 / trace|.  plane[4] = (nx, ny, nz, d) with d = -n . centroid.  Returns the number of
 * points used (0: plane / curvature are NaN).  Sign of the normal is the solver's own. */
int orc_point_normal(const float *pts, int pstride, const int32_t *idx, int64_t n, float plane[4],
                     float *curvature);

/* NormalEstimation::computeFeature (features/include/pcl/features/impl/normal_3d.hpp:46-98) over
 * queries q against the surface cloud pts: neighbours = k nearest (k > 0) or all within radius
 * (k == 0, radius > 0), ascending by (d2, index); normal flipped towards the viewpoint
 * (flipNormalTowardsViewpoint, normal_3d.h:162-181).  out = nq x (nx, ny, nz, curvature); rows
 * without neighbours or with a non-finite query are NaN.  Returns 0. */
int orc_normals(const orc_tree *t, const float *pts, int pstride, const float *q, int64_t nq,
                int qstride, int k, double radius, const float viewpoint[3], float *out,
                int nthreads);

/* ---- point-to-plane registration pieces ---- */

/* TransformationEstimationPointToPlaneLLS::estimateRigidTransformation
 * (registration/include/pcl/registration/impl/transformation_estimation_point_to_plane_lls.hpp:
 * 141-245, constructTransformationMatrix :118-139): linearised point-to-plane least squares, 6x6
 * normal equations accumulated sequentially in fp64, x = ATA^-1 ATb, T from (alpha, beta, gamma, t).
 * tn = target normals (same rows as tgt).  Pairs with a non-finite coordinate or normal are skipped
 * (the reference also tests the SOURCE normal; pass sn = NULL when the source has none). */
void orc_estimate_point_to_plane_lls(const float *src, int sstride, const float *sn, int snstride,
                                     const int32_t *sidx, const float *tgt, int tstride,
                                     const float *tn, int tnstride, const int32_t *tidx, int64_t n,
                                     float T[16]);

/* CorrespondenceEstimationNormalShooting::determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation_normal_shooting.hpp:60-165):
 * among the k nearest target points of a source point, the one closest to the line through the source
 * point along its normal (|N x V|^2, first minimum in ascending-distance order); dropped when that
 * value exceeds max_distance (as the reference compares it: not squared); distance = the squared
 * point distance of the chosen neighbour.  out must hold ns entries; returns the count. */
int64_t orc_correspondences_normal_shooting(const float *src, int64_t ns, int sstride,
                                            const float *sn, int snstride, const float *tgt,
                                            int64_t nt, int tstride, int k, double max_distance,
                                            orc_corr *out);

/* CorrespondenceEstimationBackProjection::determineCorrespondences
 * (registration/include/pcl/registration/impl/correspondence_estimation_backprojection.hpp:58-107), restated
 * with its two quirks: the weight (2 - cos^2) multiplies nn_dists[min_index] - the distance of the RUNNING
 * arg-min, not of neighbour j (:84) - and min_index is declared outside the loop over the source points and
 * never reset (:71).  No reference test covers this class: parity unpinned (restatement only).
 * out must hold ns entries; returns the count. */
int64_t orc_correspondences_back_projection(const float *src, int64_t ns, int sstride, const float *sn,
                                            int snstride, const float *tgt, int64_t nt, int tstride,
                                            const float *tn, int tnstride, int k, double max_distance,
                                            orc_corr *out);

#ifdef __cplusplus
}
#endif
#endif
