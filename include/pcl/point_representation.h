// pcl/point_representation.h - how a point becomes the k-D vector the tree indexes
// (common/include/pcl/point_representation.h:54-214).  The MI355X tree is three-dimensional:
// representations with 1..3 dimensions are supported (missing dimensions are indexed as 0, which
// contributes exactly (0 - 0)^2 = 0 to every distance); more than 3 are rejected by the trees.
#ifndef PCLMI_FACADE_POINT_REPRESENTATION_H_
#define PCLMI_FACADE_POINT_REPRESENTATION_H_

#include <cmath>
#include <memory>
#include <vector>

namespace pcl {

template <typename PointT>
class PointRepresentation {
 public:
  typedef std::shared_ptr<PointRepresentation<PointT> > Ptr;
  typedef std::shared_ptr<const PointRepresentation<PointT> > ConstPtr;

  PointRepresentation() : nr_dimensions_(0), alpha_(0), trivial_(false) {}
  virtual ~PointRepresentation() {}

  virtual void copyToFloatArray(const PointT &p, float *out) const = 0;

  bool isTrivial() const { return trivial_ && alpha_.empty(); }

  virtual bool isValid(const PointT &p) const {  // point_representation.h:96-137
    float tmp[8];
    std::vector<float> big;
    float *t = tmp;
    if (nr_dimensions_ > 8) {
      big.resize(nr_dimensions_);
      t = big.data();
    }
    copyToFloatArray(p, t);
    for (int i = 0; i < nr_dimensions_; ++i)
      if (!std::isfinite(t[i])) return false;
    return true;
  }

  template <typename OutputType>
  void vectorize(const PointT &p, OutputType &out) const {  // :141-157
    float tmp[8];
    std::vector<float> big;
    float *t = tmp;
    if (nr_dimensions_ > 8) {
      big.resize(nr_dimensions_);
      t = big.data();
    }
    copyToFloatArray(p, t);
    if (alpha_.empty())
      for (int i = 0; i < nr_dimensions_; ++i) out[i] = t[i];
    else
      for (int i = 0; i < nr_dimensions_; ++i) out[i] = t[i] * alpha_[i];
  }

  void setRescaleValues(const float *rescale_array) {  // :163-168
    alpha_.resize(nr_dimensions_);
    for (int i = 0; i < nr_dimensions_; ++i) alpha_[i] = rescale_array[i];
  }
  int getNumberOfDimensions() const { return nr_dimensions_; }

  /** \brief Not in the reference: true when vectorize() is "the point's x, y, z, each times a constant" (1..3
    * dimensions; missing ones count as 0) and then fills those constants.  That is what the device ICP loop can apply
    * to a moving source point by itself (Registration::setPointRepresentation); representations that read other
    * fields or compute something else return false. */
  virtual bool diagonalScale(float scale[3]) const {
    (void)scale;
    return false;
  }

 protected:
  bool fillScale(float scale[3]) const {
    if (nr_dimensions_ < 1 || nr_dimensions_ > 3) return false;
    for (int i = 0; i < 3; ++i) scale[i] = i < nr_dimensions_ ? (alpha_.empty() ? 1.f : alpha_[i]) : 0.f;
    return true;
  }
  int nr_dimensions_;
  std::vector<float> alpha_;
  bool trivial_;
};

// x, y, z of any point type whose first three floats are the coordinates (:188-213)
template <typename PointT>
class DefaultPointRepresentation : public PointRepresentation<PointT> {
 public:
  typedef std::shared_ptr<DefaultPointRepresentation<PointT> > Ptr;
  typedef std::shared_ptr<const DefaultPointRepresentation<PointT> > ConstPtr;
  DefaultPointRepresentation() {
    this->nr_dimensions_ = 3;
    this->trivial_ = true;
  }
  Ptr makeShared() const { return Ptr(new DefaultPointRepresentation<PointT>(*this)); }
  void copyToFloatArray(const PointT &p, float *out) const override {
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
  }
  bool diagonalScale(float scale[3]) const override { return this->fillScale(scale); }
};

// the first `max_dim` coordinates starting at `start_dim` (CustomPointRepresentation, :497-553)
template <typename PointT>
class CustomPointRepresentation : public PointRepresentation<PointT> {
 public:
  typedef std::shared_ptr<CustomPointRepresentation<PointT> > Ptr;
  typedef std::shared_ptr<const CustomPointRepresentation<PointT> > ConstPtr;
  explicit CustomPointRepresentation(int max_dim = 3, int start_dim = 0) : max_dim_(max_dim), start_dim_(start_dim) {
    const int avail = static_cast<int>(sizeof(PointT) / sizeof(float)) - start_dim_;
    this->nr_dimensions_ = avail < max_dim_ ? avail : max_dim_;
    if (this->nr_dimensions_ < 0) this->nr_dimensions_ = 0;
  }
  Ptr makeShared() const { return Ptr(new CustomPointRepresentation<PointT>(*this)); }
  void copyToFloatArray(const PointT &p, float *out) const override {
    const float *f = reinterpret_cast<const float *>(&p) + start_dim_;
    for (int i = 0; i < this->nr_dimensions_; ++i) out[i] = f[i];
  }
  bool diagonalScale(float scale[3]) const override { return start_dim_ == 0 && this->fillScale(scale); }

 protected:
  int max_dim_, start_dim_;
};

}  // namespace pcl
#endif
