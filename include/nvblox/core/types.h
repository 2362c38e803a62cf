// nvblox/core/types.h -- basic types of the nvblox:: C++ API as used by nvblox_ros (reference include sites:
// nvblox_ros/include/nvblox_ros/nvblox_node.hpp:21, layer_publishing.hpp:30-32).  The reference takes these from Eigen
// (Vector3f, Vector3i = Index3D, Isometry3f = Transform); Eigen is not a dependency of libnvblox_hip, so the types here
// are small layout-compatible stand-ins (column-major 4x4 float Transform, 12-byte vectors) that real Eigen types can
// replace by a typedef when nvblox_ros is built against this library.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace nvblox {

struct Vector3f {
  float v[3] = {0.f, 0.f, 0.f};
  Vector3f() = default;
  Vector3f(float x, float y, float z) : v{x, y, z} {}
  float x() const { return v[0]; } float y() const { return v[1]; } float z() const { return v[2]; }
  float& x() { return v[0]; } float& y() { return v[1]; } float& z() { return v[2]; }
  float operator[](int i) const { return v[i]; } float& operator[](int i) { return v[i]; }
  const float* data() const { return v; }
  Vector3f operator+(const Vector3f& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
  Vector3f operator-(const Vector3f& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
  Vector3f operator*(float s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
  float norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};

struct Vector3i {
  int32_t v[3] = {0, 0, 0};
  Vector3i() = default;
  Vector3i(int32_t x, int32_t y, int32_t z) : v{x, y, z} {}
  int32_t x() const { return v[0]; } int32_t y() const { return v[1]; } int32_t z() const { return v[2]; }
  int32_t& x() { return v[0]; } int32_t& y() { return v[1]; } int32_t& z() { return v[2]; }
  int32_t operator[](int i) const { return v[i]; } int32_t& operator[](int i) { return v[i]; }
  bool operator==(const Vector3i& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
  bool operator<(const Vector3i& o) const {
    if (v[0] != o.v[0]) return v[0] < o.v[0];
    if (v[1] != o.v[1]) return v[1] < o.v[1];
    return v[2] < o.v[2];
  }
};
using Index3D = Vector3i;
struct Vector2f {
  float v[2] = {0.f, 0.f};
  Vector2f() = default;
  Vector2f(float x, float y) : v{x, y} {}
  float x() const { return v[0]; } float y() const { return v[1]; }
};
static_assert(sizeof(Vector3f) == 12 && sizeof(Index3D) == 12, "Eigen-compatible sizes");

// nvblox_rviz_plugin/include/nvblox_rviz_plugin/nvblox_hash_utils.h:40-50 (copy of nvblox/core/hash.h in the reference)
struct Index3DHash {
  static constexpr size_t sl = 17191;
  static constexpr size_t sl2 = sl * sl;
  std::size_t operator()(const Index3D& index) const {
    return static_cast<unsigned int>(index.x() + index.y() * sl + index.z() * sl2);
  }
};

// nvblox::Plane as the node's ground-plane visualisation reads it (visualization.cpp:43-69: getHeightAtXY): n . p + d = 0
class Plane {
 public:
  Plane() = default;
  Plane(const Vector3f& normal, float d) : n_(normal), d_(d) {}
  const Vector3f& normal() const { return n_; }
  float d() const { return d_; }
  float getHeightAtXY(const Vector2f& xy) const { return n_.z() != 0.f ? -(n_.x() * xy.x() + n_.y() * xy.y() + d_) / n_.z() : 0.f; }
 private:
  Vector3f n_{0.f, 0.f, 1.f};
  float d_ = 0.f;
};

// Eigen::Isometry3f stand-in: column-major 4x4 (Eigen's default storage), rigid transform p_A = T_A_B * p_B.
class Transform {
 public:
  Transform() { setIdentity(); }
  static Transform Identity() { return Transform(); }
  void setIdentity() { std::memset(m_, 0, sizeof(m_)); m_[0] = m_[5] = m_[10] = m_[15] = 1.f; }
  float operator()(int r, int c) const { return m_[4 * c + r]; }
  float& operator()(int r, int c) { return m_[4 * c + r]; }
  const float* data() const { return m_; }   // column-major, 16 floats
  float* data() { return m_; }
  Vector3f translation() const { return {m_[12], m_[13], m_[14]}; }
  void setTranslation(const Vector3f& t) { m_[12] = t.x(); m_[13] = t.y(); m_[14] = t.z(); }
  static Transform fromRowMajor(const float* r16) { Transform t; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) t(r, c) = r16[4 * r + c]; return t; }
  void toRowMajor(float* r16) const { for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) r16[4 * r + c] = (*this)(r, c); }
  Vector3f operator*(const Vector3f& p) const {
    Vector3f o;
    for (int r = 0; r < 3; r++) o[r] = (*this)(r, 0) * p.x() + (*this)(r, 1) * p.y() + (*this)(r, 2) * p.z() + (*this)(r, 3);
    return o;
  }
  Transform operator*(const Transform& b) const {
    Transform o;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
      float s = 0.f; for (int k = 0; k < 4; k++) s += (*this)(r, k) * b(k, c); o(r, c) = s; }
    return o;
  }
  Transform inverse() const {   // rigid inverse
    Transform o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o(r, c) = (*this)(c, r);
    for (int r = 0; r < 3; r++) o(r, 3) = -(o(r, 0) * m_[12] + o(r, 1) * m_[13] + o(r, 2) * m_[14]);
    return o;
  }
 private:
  float m_[16];
};
static_assert(sizeof(Transform) == 64, "Eigen::Isometry3f-compatible size");

// nvblox::Time: integer milliseconds (MultiMapper::integrateDepth(..., update_time_ms), nvblox_node.cpp:1062)
class Time {
 public:
  Time() = default;
  explicit Time(int64_t v) : v_(v) {}
  explicit operator int64_t() const { return v_; }
  Time operator-(const Time& o) const { return Time(v_ - o.v_); }
  bool operator<(const Time& o) const { return v_ < o.v_; }
 private:
  int64_t v_ = 0;
};

// nvblox::Color is 3 x u8 (conversions/image_conversions.cpp:100-101 static_assert)
struct Color {
  uint8_t r = 0, g = 0, b = 0;
  Color() = default;
  Color(uint8_t r_, uint8_t g_, uint8_t b_) : r(r_), g(g_), b(b_) {}
  static constexpr size_t size() { return 3; }
};
static_assert(sizeof(Color) == 3, "ColorImage::ElementType::size() == 3");

enum class MemoryType { kDevice, kUnified, kHost };

class AxisAlignedBoundingBox {
 public:
  AxisAlignedBoundingBox() = default;
  AxisAlignedBoundingBox(const Vector3f& mn, const Vector3f& mx) : min_(mn), max_(mx) {}
  const Vector3f& min() const { return min_; }
  const Vector3f& max() const { return max_; }
  Vector3f& min() { return min_; }
  Vector3f& max() { return max_; }
  bool isEmpty() const { return !(min_.x() < max_.x()); }
 private:
  Vector3f min_{0.f, 0.f, 0.f}, max_{0.f, 0.f, 0.f};
};

// layer_publishing.cpp:211,348,527-529
inline Vector3f getCenterPositionFromBlockIndexAndVoxelIndex(float block_size, const Index3D& b, const Index3D& vox) {
  const float vs = block_size / 8.0f;
  return {b.x() * block_size + vox.x() * vs + vs / 2.f, b.y() * block_size + vox.y() * vs + vs / 2.f, b.z() * block_size + vox.z() * vs + vs / 2.f};
}
inline Vector3f getCenterPositionFromBlockIndex(float block_size, const Index3D& b) {
  return {(b.x() + 0.5f) * block_size, (b.y() + 0.5f) * block_size, (b.z() + 0.5f) * block_size};
}

}  // namespace nvblox
