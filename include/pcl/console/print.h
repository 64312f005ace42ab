// pcl/console/print.h - levelled printf-style logging (common/include/pcl/console/print.h:45-91).
#ifndef PCLMI_FACADE_CONSOLE_PRINT_H_
#define PCLMI_FACADE_CONSOLE_PRINT_H_

#include <cstdarg>
#include <cstdio>

namespace pcl {
namespace console {

enum VERBOSITY_LEVEL { L_ALWAYS, L_ERROR, L_WARN, L_INFO, L_DEBUG, L_VERBOSE };

inline VERBOSITY_LEVEL &verbosityRef() {
  static VERBOSITY_LEVEL level = L_INFO;
  return level;
}
inline void setVerbosityLevel(VERBOSITY_LEVEL level) { verbosityRef() = level; }
inline VERBOSITY_LEVEL getVerbosityLevel() { return verbosityRef(); }
inline bool isVerbosityLevelEnabled(VERBOSITY_LEVEL level) { return level <= verbosityRef(); }

inline void print(VERBOSITY_LEVEL level, const char *format, ...) {
  if (!isVerbosityLevelEnabled(level)) return;
  FILE *stream = (level == L_WARN || level == L_ERROR) ? stderr : stdout;
  va_list ap;
  va_start(ap, format);
  vfprintf(stream, format, ap);
  va_end(ap);
}

}  // namespace console
}  // namespace pcl

#define PCL_ALWAYS(...) pcl::console::print(pcl::console::L_ALWAYS, __VA_ARGS__)
#define PCL_ERROR(...) pcl::console::print(pcl::console::L_ERROR, __VA_ARGS__)
#define PCL_WARN(...) pcl::console::print(pcl::console::L_WARN, __VA_ARGS__)
#define PCL_INFO(...) pcl::console::print(pcl::console::L_INFO, __VA_ARGS__)
#define PCL_DEBUG(...) pcl::console::print(pcl::console::L_DEBUG, __VA_ARGS__)
#define PCL_VERBOSE(...) pcl::console::print(pcl::console::L_VERBOSE, __VA_ARGS__)
#endif
