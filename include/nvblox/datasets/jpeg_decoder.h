// nvblox/datasets/jpeg_decoder.h -- baseline / extended-sequential JPEG (ITU T.81: SOF0, SOF1 with 8-bit samples, Huffman coding) to
// interleaved 8-bit RGB: the colour frames of the Replica (results/frame%06d.jpg) and Redwood (image/%05d.jpg) datasets the fuser's
// loaders read ([U] the core's datasets/image_loader.h uses stb_image).  Grey and YCbCr (JFIF) images, 1 x 1 / 2 x 1 / 1 x 2 / 2 x 2
// chroma subsampling with the triangle ("fancy") upsampling libjpeg applies, restart intervals; an Adobe APP14 segment with transform 0
// means the three components already are RGB.  Progressive and arithmetic-coded files are refused (false).  Written from the
// standard: a separable 8 x 8 inverse DCT in float (IEEE 1180-class accuracy, so decoded samples agree with libjpeg's to about +-1),
// canonical Huffman tables decoded bit by bit through a (maxcode, valptr) table per code length.  Host-only, header-only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace nvblox {
namespace datasets {
namespace image_io {

namespace jpeg_detail {

struct Huff {
  bool present = false;
  int32_t mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  void build(const uint8_t* counts, const uint8_t* v, int n) {
    std::memcpy(vals, v, (size_t)n);
    int32_t code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      code += counts[l - 1]; k += counts[l - 1];
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int nbits = 0; bool hit_marker = false;
  BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
  // next entropy-coded byte: FF 00 is a stuffed FF; any other FF xx is a marker -- the segment ends, zeros are supplied from there on
  int next_byte() {
    if (hit_marker || p >= end) { hit_marker = true; return 0; }
    const int b = *p++;
    if (b == 0xFF) {
      if (p < end && *p == 0x00) { p++; return 0xFF; }
      p--; hit_marker = true; return 0;
    }
    return b;
  }
  int bit() {
    if (nbits == 0) { acc = (uint32_t)next_byte(); nbits = 8; }
    nbits--;
    return (int)((acc >> nbits) & 1u);
  }
  int bits(int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | bit(); return v; }
  void align() { nbits = 0; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
  int32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + (code - h.mincode[l])];
  }
  return -1;
}
inline int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

// zigzag position -> natural (row-major) index
static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// out[y][x] = 1/4 sum_u sum_v C(u) C(v) F[v][u] cos((2x+1) u pi / 16) cos((2y+1) v pi / 16), + 128, clamped: rows then columns
inline void idct8x8(const float* F, uint8_t* out, int stride) {
  static float c[8][8]; static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * (float)std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    init = true;
  }
  float tmp[64];
  for (int v = 0; v < 8; v++)
    for (int x = 0; x < 8; x++) { float s = 0.0f; for (int u = 0; u < 8; u++) s += c[x][u] * F[v * 8 + u]; tmp[v * 8 + x] = s; }
  for (int x = 0; x < 8; x++)
    for (int y = 0; y < 8; y++) {
      float s = 0.0f; for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
      int q = (int)std::lrintf(s + 128.0f);
      out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, pw = 0, ph = 0; std::vector<uint8_t> plane; };

// libjpeg's "fancy" upsampling: each output sample is 3/4 of the nearer and 1/4 of the farther input sample (per axis)
inline void upsample_h2(const uint8_t* in, int n_in, uint8_t* out, int n_out) {
  for (int i = 0; i < n_in; i++) {
    const int a = in[i], l = in[i > 0 ? i - 1 : 0], r = in[i + 1 < n_in ? i + 1 : n_in - 1];
    if (2 * i < n_out) out[2 * i] = (uint8_t)((3 * a + l + 1) >> 2);
    if (2 * i + 1 < n_out) out[2 * i + 1] = (uint8_t)((3 * a + r + 2) >> 2);
  }
}

}  // namespace jpeg_detail

// rgb: rows * cols * 3 bytes (grey images are replicated into the three channels); false on anything this decoder does not cover
inline bool decodeJpeg(const std::vector<uint8_t>& file, int* rows_out, int* cols_out, std::vector<uint8_t>* rgb) {
  using namespace jpeg_detail;
  const uint8_t* d = file.data(); const size_t n = file.size();
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
  float qt[4][64]; bool have_qt[4] = {false, false, false, false};
  Huff hdc[4], hac[4];
  std::vector<Component> comp;
  int W = 0, H = 0, restart = 0, adobe_transform = -1;
  size_t pos = 2;
  while (pos + 4 <= n) {
    if (d[pos] != 0xFF) return false;
    while (pos < n && d[pos] == 0xFF) pos++;          // fill bytes
    if (pos >= n) return false;
    const int marker = d[pos++];
    if (marker == 0xD9) return false;                 // EOI before a scan
    if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
    if (pos + 2 > n) return false;
    const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
    if (len < 2 || pos + len > n) return false;
    const uint8_t* s = d + pos + 2; const size_t sl = len - 2;
    if (marker == 0xDB) {                                                  // DQT
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15; o++;
        if (tq > 3 || o + (pq ? 128u : 64u) > sl) return false;
        for (int i = 0; i < 64; i++) { const int q = pq ? ((s[o + 2 * i] << 8) | s[o + 2 * i + 1]) : s[o + i]; qt[tq][kZigzag[i]] = (float)q; }
        o += pq ? 128 : 64; have_qt[tq] = true;
      }
    } else if (marker == 0xC4) {                                           // DHT
      size_t o = 0;
      while (o + 17 <= sl) {
        const int tc = s[o] >> 4, th = s[o] & 15;
        int total = 0; for (int i = 0; i < 16; i++) total += s[o + 1 + i];
        if (th > 3 || tc > 1 || total > 256 || o + 17 + (size_t)total > sl) return false;
        (tc ? hac[th] : hdc[th]).build(s + o + 1, s + o + 17, total);
        o += 17 + (size_t)total;
      }
    } else if (marker == 0xC0 || marker == 0xC1) {                         // SOF0 / SOF1 (Huffman, sequential)
      if (sl < 6 || s[0] != 8) return false;
      H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
      const int nc = s[5];
      if ((nc != 1 && nc != 3) || sl < 6 + 3 * (size_t)nc || W <= 0 || H <= 0 || (int64_t)W * H > (int64_t)1 << 28) return false;   // (a corrupt header must not allocate gigabytes)
      comp.resize((size_t)nc);
      for (int i = 0; i < nc; i++) { comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i]; if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) return false; }
    } else if (marker == 0xC2 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
      return false;                                                        // progressive / lossless / arithmetic
    } else if (marker == 0xDD) {
      if (sl < 2) return false;
      restart = (s[0] << 8) | s[1];
    } else if (marker == 0xEE) {                                           // Adobe APP14
      if (sl >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11];
    } else if (marker == 0xDA) {                                           // SOS: one interleaved scan with every component
      if (comp.empty() || sl < 1 || s[0] != (int)comp.size() || sl < 1 + 2 * comp.size() + 3) return false;
      for (size_t i = 0; i < comp.size(); i++) {
        size_t k = 0; while (k < comp.size() && comp[k].id != s[1 + 2 * i]) k++;
        if (k == comp.size()) return false;
        comp[k].td = s[2 + 2 * i] >> 4; comp[k].ta = s[2 + 2 * i] & 15;
        if (comp[k].td > 3 || comp[k].ta > 3 || !hdc[comp[k].td].present || !hac[comp[k].ta].present || !have_qt[comp[k].tq]) return false;
      }
      pos += len;
      break;
    }
    pos += len;
  }
  if (comp.empty() || pos >= n) return false;
  int hmax = 1, vmax = 1;
  for (auto& c : comp) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
  if (comp.size() == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }       // a single-component scan is not interleaved: one block per MCU
  const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
  const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
  for (auto& c : comp) { c.pw = mcus_x * c.h * 8; c.ph = mcus_y * c.v * 8; c.plane.assign((size_t)c.pw * c.ph, 0); c.pred = 0; }
  BitReader br(d + pos, d + n);
  int until_restart = restart;
  float blk[64];
  for (int my = 0; my < mcus_y; my++) for (int mx = 0; mx < mcus_x; mx++) {
    if (restart && until_restart == 0) {
      // RSTn: byte-align, skip the marker, reset the DC predictions
      br.align();
      const uint8_t* q = br.p;
      while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
      if (q + 1 >= br.end) return false;
      br.p = q + 2; br.hit_marker = false;
      for (auto& c : comp) c.pred = 0;
      until_restart = restart;
    }
    for (auto& c : comp) for (int by = 0; by < c.v; by++) for (int bx = 0; bx < c.h; bx++) {
      std::memset(blk, 0, sizeof(blk));
      const int t = decode_symbol(br, hdc[c.td]);
      if (t < 0 || t > 11) return false;
      c.pred += extend(br.bits(t), t);
      blk[0] = (float)c.pred * qt[c.tq][0];
      for (int k = 1; k < 64;) {
        const int rs = decode_symbol(br, hac[c.ta]);
        if (rs < 0) return false;
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) { if (r == 15) { k += 16; continue; } break; }       // ZRL / EOB
        k += r;
        if (k > 63) return false;
        blk[kZigzag[k]] = (float)extend(br.bits(sz), sz) * qt[c.tq][kZigzag[k]];
        k++;
      }
      idct8x8(blk, &c.plane[(size_t)((my * c.v + by) * 8) * c.pw + (size_t)(mx * c.h + bx) * 8], c.pw);
    }
    if (restart) until_restart--;
  }
  // upsample the subsampled components to the luma grid (rows first, then columns), then convert
  const int FW = mcus_x * mcu_w, FH = mcus_y * mcu_h;
  std::vector<std::vector<uint8_t>> full(comp.size());
  for (size_t i = 0; i < comp.size(); i++) {
    Component& c = comp[i];
    if (c.pw == FW && c.ph == FH) { full[i].swap(c.plane); continue; }
    // (the filter sees the component's TRUE extent, ceil(W h / hmax) x ceil(H v / vmax), with its last sample replicated -- not the
    // padding the encoder put into the rest of the last MCU)
    const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax;
    std::vector<uint8_t> wide((size_t)FW * c.ph);
    for (int r = 0; r < c.ph; r++) {
      if (c.pw == FW) std::memcpy(&wide[(size_t)r * FW], &c.plane[(size_t)r * c.pw], (size_t)FW);
      else upsample_h2(&c.plane[(size_t)r * c.pw], cw, &wide[(size_t)r * FW], FW);
    }
    if (c.ph == FH) { full[i].swap(wide); continue; }
    full[i].assign((size_t)FW * FH, 0);
    std::vector<uint8_t> col((size_t)c.ph), colo((size_t)FH);
    for (int x = 0; x < FW; x++) {
      for (int r = 0; r < c.ph; r++) col[(size_t)r] = wide[(size_t)r * FW + x];
      upsample_h2(col.data(), chh, colo.data(), FH);
      for (int r = 0; r < FH; r++) full[i][(size_t)r * FW + x] = colo[(size_t)r];
    }
  }
  rgb->resize((size_t)W * H * 3);
  const bool ycc = comp.size() == 3 && adobe_transform != 0;
  for (int r = 0; r < H; r++) for (int x = 0; x < W; x++) {
    uint8_t* o = &(*rgb)[((size_t)r * W + x) * 3];
    const size_t q = (size_t)r * FW + x;
    if (comp.size() == 1) { o[0] = o[1] = o[2] = full[0][q]; continue; }
    if (!ycc) { o[0] = full[0][q]; o[1] = full[1][q]; o[2] = full[2][q]; continue; }
    const float Y = full[0][q], cb = (float)full[1][q] - 128.0f, cr = (float)full[2][q] - 128.0f;
    const int R = (int)std::lrintf(Y + 1.402f * cr), G = (int)std::lrintf(Y - 0.344136f * cb - 0.714136f * cr), B = (int)std::lrintf(Y + 1.772f * cb);
    o[0] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R)); o[1] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G)); o[2] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
  }
  *rows_out = H; *cols_out = W;
  return true;
}

}  // namespace image_io
}  // namespace datasets
}  // namespace nvblox
