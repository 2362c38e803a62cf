// nvblox/datasets/image_loader.h -- the image decoding the dataset loaders need: PNG (8 / 16 bit, grey / RGB / RGBA, non-interlaced)
// through zlib, binary PGM / PPM, and baseline JPEG (jpeg_decoder.h: the colour frames of Replica and Redwood).  [U] the core's
// datasets/image_loader.h uses stb_image; the formats the three supported datasets ship depth in are all 16-bit PNG.  Progressive
// JPEGs are refused -- a loader then runs that frame depth-only.  Host-only, header-only; link with -lz.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "nvblox/sensors/image.h"
#include "nvblox/datasets/jpeg_decoder.h"

namespace nvblox {
namespace datasets {
namespace image_io {

struct DecodedImage { int rows = 0, cols = 0, channels = 0, bit_depth = 0; std::vector<uint16_t> data; };   // samples widened to u16, interleaved

inline bool readFile(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  if (n <= 0) { std::fclose(f); return false; }
  out->resize((size_t)n);
  const bool ok = std::fread(out->data(), 1, (size_t)n, f) == (size_t)n;
  std::fclose(f);
  return ok;
}
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool decodePng(const std::vector<uint8_t>& file, DecodedImage* img) {
  static const uint8_t kSig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), kSig, 8) != 0) return false;
  size_t pos = 8;
  int color_type = -1, interlace = 0;
  std::vector<uint8_t> idat;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
    if ((uint64_t)pos + 12 + len > file.size()) return false;
    const uint8_t* d = &file[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) return false;
      const uint32_t wc = be32(d), hr = be32(d + 4);
      if (wc == 0 || hr == 0 || wc > 32768u || hr > 32768u || (uint64_t)wc * hr > (1ull << 28)) return false;     // (the JPEG path's cap: no multi-GB allocation from a header)
      img->cols = (int)wc; img->rows = (int)hr; img->bit_depth = d[8]; color_type = d[9]; interlace = d[12];
    } else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (interlace != 0 || (img->bit_depth != 8 && img->bit_depth != 16) || img->rows <= 0 || img->cols <= 0) return false;
  img->channels = color_type == 0 ? 1 : (color_type == 2 ? 3 : (color_type == 4 ? 2 : (color_type == 6 ? 4 : 0)));
  if (!img->channels) return false;                      // (palette images are not used by the datasets)
  const int bps = img->bit_depth / 8, bpp = bps * img->channels;
  const size_t stride = (size_t)img->cols * bpp;
  std::vector<uint8_t> raw((stride + 1) * (size_t)img->rows);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
  std::vector<uint8_t> cur(stride), prev(stride, 0);
  img->data.resize((size_t)img->rows * img->cols * img->channels);
  for (int r = 0; r < img->rows; r++) {
    const uint8_t* line = &raw[(stride + 1) * (size_t)r];
    const int filter = line[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int x = line[1 + i];
      switch (filter) {
        case 0: break;
        case 1: x += a; break;
        case 2: x += b; break;
        case 3: x += (a + b) >> 1; break;
        case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); x += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: return false;
      }
      cur[i] = (uint8_t)x;
    }
    uint16_t* o = &img->data[(size_t)r * img->cols * img->channels];
    for (size_t s = 0; s < (size_t)img->cols * img->channels; s++) o[s] = bps == 2 ? (uint16_t)((cur[2 * s] << 8) | cur[2 * s + 1]) : cur[s];
    prev.swap(cur);
  }
  return true;
}
// binary PGM (P5) / PPM (P6), maxval <= 65535
inline bool decodePnm(const std::vector<uint8_t>& file, DecodedImage* img) {
  if (file.size() < 7 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) return false;
  size_t pos = 2; long v[3]; int got = 0;
  while (got < 3 && pos < file.size()) {
    while (pos < file.size() && (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t')) pos++;
    if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
    long x = 0; bool any = false;
    while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { x = x * 10 + (file[pos] - '0'); pos++; any = true; if (x > 65535) return false; }
    if (!any) return false;
    v[got++] = x;
  }
  pos++;    // one whitespace after maxval
  if (got < 3 || v[0] <= 0 || v[1] <= 0 || v[0] > 32768 || v[1] > 32768 || (uint64_t)v[0] * (uint64_t)v[1] > (1ull << 28) || v[2] <= 0) return false;
  img->cols = (int)v[0]; img->rows = (int)v[1]; img->channels = file[1] == '5' ? 1 : 3; img->bit_depth = v[2] > 255 ? 16 : 8;
  const size_t n = (size_t)img->rows * img->cols * img->channels, bps = img->bit_depth / 8;
  if (pos > file.size() || n * bps > file.size() - pos) return false;
  img->data.resize(n);
  for (size_t i = 0; i < n; i++) img->data[i] = bps == 2 ? (uint16_t)((file[pos + 2 * i] << 8) | file[pos + 2 * i + 1]) : file[pos + i];
  return true;
}
inline bool decode(const std::string& path, DecodedImage* img) {
  std::vector<uint8_t> file;
  if (!readFile(path, &file)) return false;
  if (decodePng(file, img) || decodePnm(file, img)) return true;
  std::vector<uint8_t> rgb;
  if (!decodeJpeg(file, &img->rows, &img->cols, &rgb)) return false;
  img->channels = 3; img->bit_depth = 8;
  img->data.assign(rgb.begin(), rgb.end());
  return true;
}

}  // namespace image_io

// [U] datasets::load16BitDepthImage / load8BitColorImage: depth = raw * scaling_factor (metres), invalid stays 0
inline bool load16BitDepthImage(const std::string& path, DepthImage* depth, float scaling_factor, const CudaStream& stream, std::vector<float>* host_scratch) {
  image_io::DecodedImage img;
  if (!image_io::decode(path, &img) || img.channels != 1) return false;
  host_scratch->resize(img.data.size());
  for (size_t i = 0; i < img.data.size(); i++) (*host_scratch)[i] = (float)img.data[i] * scaling_factor;
  depth->copyFromAsync(img.rows, img.cols, host_scratch->data(), stream);
  stream.synchronize();                              // (the scratch vector is reused by the next frame)
  return true;
}
inline bool load8BitColorImage(const std::string& path, ColorImage* color, const CudaStream& stream, std::vector<Color>* host_scratch) {
  image_io::DecodedImage img;
  if (!image_io::decode(path, &img) || img.channels < 3 || img.bit_depth != 8) return false;
  host_scratch->resize((size_t)img.rows * img.cols);
  for (size_t i = 0; i < host_scratch->size(); i++)
    (*host_scratch)[i] = Color((uint8_t)img.data[i * img.channels], (uint8_t)img.data[i * img.channels + 1], (uint8_t)img.data[i * img.channels + 2]);
  color->copyFromAsync(img.rows, img.cols, host_scratch->data(), stream);
  stream.synchronize();
  return true;
}

}  // namespace datasets
}  // namespace nvblox
