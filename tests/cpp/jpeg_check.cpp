// jpeg_check <in.jpg> <out.ppm>: decodes with nvblox/datasets/jpeg_decoder.h and writes a binary PPM (tests/test_cpp_facade.py compares it
// with PIL's decode of the same file).  Exit code 2 = the decoder refused the file.
#include <cstdio>
#include <vector>
#include "nvblox/datasets/jpeg_decoder.h"

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> file((size_t)n);
  if (std::fread(file.data(), 1, (size_t)n, f) != (size_t)n) return 1;
  std::fclose(f);
  int rows = 0, cols = 0; std::vector<uint8_t> rgb;
  if (!nvblox::datasets::image_io::decodeJpeg(file, &rows, &cols, &rgb)) return 2;
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 1;
  std::fprintf(o, "P6\n%d %d\n255\n", cols, rows);
  std::fwrite(rgb.data(), 1, rgb.size(), o);
  std::fclose(o);
  return 0;
}
