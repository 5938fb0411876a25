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

// {fmt}'s "{}" of a double (what the reference prints ratios, MAE and MSE with): the shortest digits
// that round-trip, laid out in fixed notation while the decimal exponent is in [-4, 16) and as
// d.ddde[+-]XX otherwise.  std::to_chars alone would pick whichever of the two is shorter ("1e-04"
// for 0.0001, "1e+05" for 100000), so only its digits and exponent are taken from it.
std::string num(double v)
{
  if (std::isnan(v)) return std::signbit(v) ? "-nan" : "nan";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  if (v == 0.0) return std::signbit(v) ? "-0" : "0";
  char buf[64];
  auto res = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
  std::string sci(buf, res.ptr);                      // [-]d[.ddd]e[+-]XX
  const bool neg = sci[0] == '-';
  if (neg) sci.erase(0, 1);
  const size_t epos = sci.find('e');
  std::string digits = sci.substr(0, epos);
  const int exp10 = std::atoi(sci.c_str() + epos + 1);
  const size_t dot = digits.find('.');
  if (dot != std::string::npos) digits.erase(dot, 1);  // all significant digits, first one has weight 10^exp10
  std::string out;
  if (exp10 >= -4 && exp10 < 16) {
    if (exp10 < 0) {
      out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    } else if ((size_t)exp10 + 1 >= digits.size()) {
      out = digits + std::string((size_t)exp10 + 1 - digits.size(), '0');
    } else {
      out = digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    }
  } else {
    out = digits.substr(0, 1);
    if (digits.size() > 1) out += "." + digits.substr(1);
    char e[16];
    std::snprintf(e, sizeof(e), "e%c%02d", exp10 < 0 ? '-' : '+', exp10 < 0 ? -exp10 : exp10);
    out += e;
  }
  return neg ? "-" + out : out;
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
