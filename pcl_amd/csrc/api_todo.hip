// TEMPORARY: entry points still to be implemented fail loudly.
#include "pclmi_internal.h"
#define TODO(name) pclmi::set_error(name ": not implemented yet"); return PCLMI_ERR_UNSUPPORTED
extern "C" {
int pclmi_radius(const pclmi_tree *, const float *, size_t, size_t, double, uint32_t, int, uint64_t *, int32_t **, float **) { TODO("pclmi_radius"); }
int pclmi_radius_dev(const pclmi_tree *, const float *, size_t, size_t, double, uint32_t, int, uint64_t *, int32_t **, float **, uint64_t *, void *) { TODO("pclmi_radius_dev"); }
}
