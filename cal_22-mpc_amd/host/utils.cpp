#include "utils.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>

bool isFileExists(const std::string &filePath)
{
  std::ifstream f(filePath);
  return f.good();
}

std::string parseConfig(const std::string &configPath)
{
  std::vector<std::string> parts = mpctext::split(configPath, "/");
  std::string name = parts[parts.size() - 1];
  if (!mpctext::replace_all(name, ".json", "")) {
    printf("String parsing have failed!\n");
    exit(1);
  }
  return name;
}

namespace mpctext
{

std::string num(double v)
{
  if (std::isnan(v)) return std::signbit(v) ? "-nan" : "nan";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  char buf[64];
  auto res = std::to_chars(buf, buf + sizeof(buf), v);
  return std::string(buf, res.ptr);
}

std::string num(uint64_t v) { return std::to_string(v); }

std::vector<std::string> split(const std::string &s, const std::string &delim)
{
  std::vector<std::string> out;
  size_t pos = 0;
  for (;;) {
    size_t n = s.find(delim, pos);
    if (n == std::string::npos) {
      out.push_back(s.substr(pos));
      break;
    }
    out.push_back(s.substr(pos, n - pos));
    pos = n + delim.size();
  }
  return out;
}

bool ends_with(const std::string &s, const std::string &suffix)
{
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

bool replace_all(std::string &s, const std::string &from, const std::string &to)
{
  bool any = false;
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.size(), to);
    pos += to.size();
    any = true;
  }
  return any;
}

}  // namespace mpctext
