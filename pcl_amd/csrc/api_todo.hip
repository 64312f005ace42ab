// TEMPORARY: entry points still to be implemented fail loudly.
#include "pclmi_internal.h"
#define TODO(name) pclmi::set_error(name ": not implemented yet"); return PCLMI_ERR_UNSUPPORTED
extern "C" {
int pclmi_radius(const pclmi_tree *, const float *, size_t, size_t, double, uint32_t, int, uint64_t *, int32_t **, float **) { TODO("pclmi_radius"); }
int pclmi_radius_dev(const pclmi_tree *, const float *, size_t, size_t, double, uint32_t, int, uint64_t *, int32_t **, float **, uint64_t *, void *) { TODO("pclmi_radius_dev"); }
int pclmi_correspondences(const pclmi_tree *, const float *, size_t, size_t, const int32_t *, size_t, double, pclmi_correspondence *, size_t *) { TODO("pclmi_correspondences"); }
int pclmi_correspondences_reciprocal(const pclmi_tree *, const pclmi_tree *, const float *, size_t, size_t, const int32_t *, size_t, const float *, size_t, size_t, double, pclmi_correspondence *, size_t *) { TODO("x"); }
int pclmi_estimate_rigid_svd(const float *, size_t, size_t, const int32_t *, const float *, size_t, size_t, const int32_t *, size_t, float[16]) { TODO("x"); }
int pclmi_transform(float *, size_t, size_t, const float[16]) { TODO("x"); }
void pclmi_icp_default_params(pclmi_icp_params *) {}
int pclmi_icp_align(const pclmi_icp_params *, const pclmi_tree *, const float *, size_t, size_t, const float *, float[16], float *, pclmi_icp_stats *) { TODO("x"); }
int pclmi_fitness_score(const pclmi_tree *, const float *, size_t, size_t, const float[16], double, double *) { TODO("x"); }
int pclmi_icp_begin(const pclmi_tree *, const float *, size_t, size_t, const float *, pclmi_icp_ctx **) { TODO("x"); }
int pclmi_icp_step_dev(pclmi_icp_ctx *, double, double *, void *) { TODO("x"); }
int pclmi_icp_step(pclmi_icp_ctx *, double, double[PCLMI_ICP_NSUMS]) { TODO("x"); }
int pclmi_icp_solve(const double[PCLMI_ICP_NSUMS], float[16]) { TODO("x"); }
int pclmi_icp_set_transform(pclmi_icp_ctx *, const float[16]) { TODO("x"); }
int pclmi_icp_end(pclmi_icp_ctx *, float *) { TODO("x"); }
}
