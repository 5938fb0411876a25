// utils.h -- small helpers of the CLI (reference src/utils.{h,cpp}) plus the text
// formatting the reference gets from {fmt}: "{}" of a double is the shortest
// round-trip representation in fixed notation for decimal exponents -4 .. 15,
// exponent notation otherwise (mpctext::num).
#ifndef MPC_HOST_UTILS_H
#define MPC_HOST_UTILS_H

#include <cstdint>
#include <string>
#include <vector>

bool isFileExists(const std::string &filePath);
// config path -> file name without ".json" (the stem of the result CSV files)
std::string parseConfig(const std::string &configPath);

namespace mpctext
{
std::string num(double v);
std::string num(uint64_t v);
std::vector<std::string> split(const std::string &s, const std::string &delim);
bool ends_with(const std::string &s, const std::string &suffix);
bool replace_all(std::string &s, const std::string &from, const std::string &to);
}  // namespace mpctext

#endif
