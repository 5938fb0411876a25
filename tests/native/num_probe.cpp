// num_probe.cpp -- CPU test driver: mpctext::num(double) (the text the CSV writers print doubles with)
// for hexadecimal-float arguments, one result per line.
#include <cstdio>
#include <cstdlib>

#include "utils.h"

int main(int argc, char **argv)
{
  for (int i = 1; i < argc; i++) std::printf("%s\n", mpctext::num(std::strtod(argv[i], nullptr)).c_str());
  return 0;
}
