// pcl/console/parse.h - the command-line helpers tools/icp.cpp uses
// (common/include/pcl/console/parse.h: parse_argument, parse_file_extension_argument).
#ifndef PCLMI_FACADE_CONSOLE_PARSE_H_
#define PCLMI_FACADE_CONSOLE_PARSE_H_

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pcl {
namespace console {

/** \brief index of the value following `str` on the command line, or -1 */
inline int find_argument_value(int argc, char **argv, const char *str) {
  for (int i = 1; i < argc - 1; ++i)
    if (std::strcmp(argv[i], str) == 0) return i + 1;
  return -1;
}
inline bool find_switch(int argc, char **argv, const char *str) {
  for (int i = 1; i < argc; ++i)
    if (std::strcmp(argv[i], str) == 0) return true;
  return false;
}
inline int parse_argument(int argc, char **argv, const char *str, double &val) {
  const int i = find_argument_value(argc, argv, str);
  if (i > 0) val = std::atof(argv[i]);
  return i > 0 ? i - 1 : -1;
}
inline int parse_argument(int argc, char **argv, const char *str, float &val) {
  const int i = find_argument_value(argc, argv, str);
  if (i > 0) val = static_cast<float>(std::atof(argv[i]));
  return i > 0 ? i - 1 : -1;
}
inline int parse_argument(int argc, char **argv, const char *str, int &val) {
  const int i = find_argument_value(argc, argv, str);
  if (i > 0) val = std::atoi(argv[i]);
  return i > 0 ? i - 1 : -1;
}
inline int parse_argument(int argc, char **argv, const char *str, bool &val) {
  const int i = find_argument_value(argc, argv, str);
  if (i > 0) val = std::atoi(argv[i]) == 1;
  return i > 0 ? i - 1 : -1;
}
inline int parse_argument(int argc, char **argv, const char *str, std::string &val) {
  const int i = find_argument_value(argc, argv, str);
  if (i > 0) val = argv[i];
  return i > 0 ? i - 1 : -1;
}
/** \brief indices of the arguments ending in `ext` (e.g. ".pcd") */
inline std::vector<int> parse_file_extension_argument(int argc, char **argv, const std::string &ext) {
  std::vector<int> idx;
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a.size() > ext.size() && a[0] != '-' && a.compare(a.size() - ext.size(), ext.size(), ext) == 0)
      idx.push_back(i);
  }
  return idx;
}

}  // namespace console
}  // namespace pcl
#endif
