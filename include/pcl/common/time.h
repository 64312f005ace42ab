// pcl/common/time.h - StopWatch / ScopeTime (common/include/pcl/common/time.h:58-133) on std::chrono.
// For GPU kernel time use pclmi_timing_enable / pclmi_timing_read (include/pclmi.h).
#ifndef PCLMI_FACADE_COMMON_TIME_H_
#define PCLMI_FACADE_COMMON_TIME_H_

#include <chrono>
#include <iostream>
#include <string>

namespace pcl {

class StopWatch {
 public:
  StopWatch() : start_(std::chrono::steady_clock::now()) {}
  virtual ~StopWatch() {}
  double getTime() const {  // milliseconds
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start_).count();
  }
  double getTimeSeconds() const { return getTime() * 0.001; }
  void reset() { start_ = std::chrono::steady_clock::now(); }

 protected:
  std::chrono::steady_clock::time_point start_;
};

class ScopeTime : public StopWatch {
 public:
  explicit ScopeTime(const char *title) : title_(title) {}
  ScopeTime() : title_("") {}
  ~ScopeTime() { std::cerr << title_ << " took " << getTime() << "ms.\n"; }

 private:
  std::string title_;
};

inline double getTime() {
  return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

}  // namespace pcl
#endif
