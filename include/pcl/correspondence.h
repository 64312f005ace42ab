// pcl/correspondence.h - pcl::Correspondence (common/include/pcl/correspondence.h:53-92).
// Binary-compatible with pclmi_correspondence (include/pclmi.h): {int, int, float}.
#ifndef PCLMI_FACADE_CORRESPONDENCE_H_
#define PCLMI_FACADE_CORRESPONDENCE_H_

#include <limits>
#include <memory>
#include <vector>

#include <pclmi.h>

namespace pcl {

struct Correspondence {
  int index_query;
  int index_match;
  union {
    float distance;
    float weight;
  };
  Correspondence() : index_query(0), index_match(-1), distance(std::numeric_limits<float>::max()) {}
  Correspondence(int q, int m, float d) : index_query(q), index_match(m), distance(d) {}
};
static_assert(sizeof(Correspondence) == sizeof(pclmi_correspondence), "Correspondence must match the C ABI record");

typedef std::vector<Correspondence> Correspondences;
typedef std::shared_ptr<Correspondences> CorrespondencesPtr;
typedef std::shared_ptr<const Correspondences> CorrespondencesConstPtr;

}  // namespace pcl
#endif
