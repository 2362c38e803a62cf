// nvblox/sensors/camera.h -- pinhole camera as built at conversions/image_conversions.cpp:27-32:
// Camera(K[0], K[4], K[2], K[5], width, height).
#pragma once
#include "nvblox/core/types.h"
#include "nvblox_hip.h"

namespace nvblox {

class Camera {
 public:
  Camera() = default;
  Camera(float fu, float fv, float cu, float cv, int width, int height) : c_{fu, fv, cu, cv, width, height} {}
  float fu() const { return c_.fu; } float fv() const { return c_.fv; }
  float cu() const { return c_.cu; } float cv() const { return c_.cv; }
  int width() const { return c_.width; } int height() const { return c_.height; }
  int cols() const { return c_.width; } int rows() const { return c_.height; }
  const nvbx_camera& c_abi() const { return c_; }
 private:
  nvbx_camera c_{0.f, 0.f, 0.f, 0.f, 0, 0};
};

}  // namespace nvblox
