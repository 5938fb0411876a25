#include "CompResult.h"

namespace comp
{

void CompResult::Update(unsigned uncompSize, unsigned compSize, int selected)
{
  (void)selected;
  OriginalSize += uncompSize;
  CompressedSize += compSize;
  CompRatio = (double)OriginalSize / (double)CompressedSize;      // reference CompResult.h:30-35
}

void CompResult::openForAppend(std::ofstream &file, const std::string &filePath, const std::string &header)
{
  if (!isFileExists(filePath)) {
    file.open(filePath);
    if (!file.is_open()) {
      std::cout << "File is not open: \"" << filePath << "\"" << std::endl;
      exit(1);
    }
    file << header;
    file.close();
  }
  file.open(filePath, std::ios_base::app);
}

void CompResult::Print(std::string workloadName, std::string filePath)
{
  std::ofstream file;
  if (filePath != "") openForAppend(file, filePath, "workload,original_size,compressed_size,compression_ratio,\n");
  std::ostream &stream = (filePath == "") ? std::cout : file;
  stream << workloadName << "," << OriginalSize << "," << CompressedSize << "," << mpctext::num(CompRatio) << ","
         << std::endl;
}

void CompResult::PrintDetail(std::string workloadName, std::string filePath)
{
  (void)workloadName;
  (void)filePath;
}

}  // namespace comp
