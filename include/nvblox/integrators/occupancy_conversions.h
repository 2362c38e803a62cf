// nvblox/integrators/occupancy_conversions.h -- conversions::saveOccupancyGridAsPng / saveOccupancyGridYaml: the costmap the node
// writes at shutdown for Nav2's map_server (nvblox_node.cpp:140-168; after_shutdown_map_save_path).  The grid is the
// int8 image of EsdfSlicer::occupancyGridFromSliceImage (100 occupied, 0 free, -1 unknown; row = y, col = x, origin at the
// AABB's minimum corner).  [U] The reference's writers live in the absent core; this follows the map_server file format the
// thresholds are named after: 8-bit grey PNG with row 0 = the map's TOP (maximum y), pixel = 254 free / 0 occupied / 205
// unknown, and a YAML with image, mode trinary, resolution, origin [x, y, 0], negate 0, occupied_thresh, free_thresh
// (map_server reads p = (255 - pixel) / 255: 0 -> 1.0 >= occupied_thresh, 254 -> 0.004 <= free_thresh, 205 -> 0.196 in between
// = unknown, for the node's 0.65 / 0.25).  The PNG is written without zlib: stored (uncompressed) deflate blocks.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace nvblox {
namespace conversions {

namespace detail {
inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
  static uint32_t table[256]; static bool init = false;
  if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : (c >> 1); table[i] = c; } init = true; }
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}
inline void put32(std::vector<uint8_t>* v, uint32_t x) { v->push_back((uint8_t)(x >> 24)); v->push_back((uint8_t)(x >> 16)); v->push_back((uint8_t)(x >> 8)); v->push_back((uint8_t)x); }
inline void chunk(std::vector<uint8_t>* out, const char type[4], const std::vector<uint8_t>& data) {
  put32(out, (uint32_t)data.size());
  std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
  out->insert(out->end(), td.begin(), td.end());
  put32(out, crc32(td.data(), td.size()));
}
// 8-bit greyscale PNG, rows top to bottom
inline std::vector<uint8_t> encodeGreyPng(const uint8_t* pixels, uint32_t width, uint32_t height) {
  std::vector<uint8_t> raw; raw.reserve((size_t)(width + 1) * height);
  for (uint32_t r = 0; r < height; r++) { raw.push_back(0); raw.insert(raw.end(), pixels + (size_t)r * width, pixels + (size_t)(r + 1) * width); }   // filter type 0
  std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);                          // zlib header, no compression
  uint32_t a = 1, b = 0;                                                                 // Adler-32 of the raw stream
  for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
  size_t pos = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n >= raw.size() ? 1 : 0);                                          // BFINAL, BTYPE = 00 (stored)
    z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  put32(&z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr; put32(&ihdr, width); put32(&ihdr, height);
  ihdr.push_back(8); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bit, grey, deflate, filter 0, no interlace
  chunk(&out, "IHDR", ihdr); chunk(&out, "IDAT", z); chunk(&out, "IEND", {});
  return out;
}
}  // namespace detail

// nvblox_node.cpp:156-158: (path, free_threshold, occupied_threshold, height, width, grid)
inline bool saveOccupancyGridAsPng(const std::string& path, float /*free_threshold*/, float /*occupied_threshold*/, size_t height, size_t width,
                                   const std::vector<int8_t>& occupancy_grid) {
  if (height == 0 || width == 0 || occupancy_grid.size() < height * width) return false;
  std::vector<uint8_t> px(height * width);
  for (size_t r = 0; r < height; r++) for (size_t c = 0; c < width; c++) {
    const int8_t o = occupancy_grid[(height - 1 - r) * width + c];                       // image row 0 = top of the map = largest y
    px[r * width + c] = o < 0 ? 205 : (o >= 50 ? 0 : 254);
  }
  const std::vector<uint8_t> png = detail::encodeGreyPng(px.data(), (uint32_t)width, (uint32_t)height);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
  return (std::fclose(f) == 0) && ok;
}

// nvblox_node.cpp:164-166: (path, image_name, voxel_size, origin_x, origin_y, free_threshold, occupied_threshold)
inline bool saveOccupancyGridYaml(const std::string& path, const std::string& image_name, float voxel_size, float origin_x, float origin_y,
                                  float free_threshold, float occupied_threshold) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  const bool ok = std::fprintf(f, "image: %s\nmode: trinary\nresolution: %.6g\norigin: [%.6g, %.6g, 0.0]\nnegate: 0\noccupied_thresh: %.6g\nfree_thresh: %.6g\n",
                               image_name.c_str(), voxel_size, origin_x, origin_y, occupied_threshold, free_threshold) > 0;
  return (std::fclose(f) == 0) && ok;
}

}  // namespace conversions
}  // namespace nvblox
