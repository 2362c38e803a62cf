// nvblox/sensors/lidar.h -- spinning-LiDAR intrinsics as constructed at nvblox_node.cpp:1315-1323:
//   Lidar(lidar_width, lidar_height, lidar_min_valid_range_m, lidar_vertical_fov_rad)                    (beams centred on 0 elevation)
//   Lidar(lidar_width, lidar_height, lidar_min_valid_range_m, min_angle_below_zero_elevation_rad, max_angle_above_zero_elevation_rad)
// The second form takes the magnitudes of the lowest / highest beam elevation (node_params.hpp:138-146 documents positive
// numbers; nvblox_os1.yaml:11 passes the lower one negative) -- the sign of the lower angle is therefore ignored.
#pragma once
#include <cmath>
#include "nvblox_hip.h"

namespace nvblox {

class Lidar {
 public:
  Lidar() = default;
  Lidar(int num_azimuth_divisions, int num_elevation_divisions, float min_valid_range_m, float vertical_fov_rad)
      : l_{num_azimuth_divisions, num_elevation_divisions, min_valid_range_m, -0.5f * vertical_fov_rad, 0.5f * vertical_fov_rad} {}
  Lidar(int num_azimuth_divisions, int num_elevation_divisions, float min_valid_range_m, float min_angle_below_zero_elevation_rad,
        float max_angle_above_zero_elevation_rad)
      : l_{num_azimuth_divisions, num_elevation_divisions, min_valid_range_m, -std::fabs(min_angle_below_zero_elevation_rad),
           std::fabs(max_angle_above_zero_elevation_rad)} {}
  int num_azimuth_divisions() const { return l_.num_azimuth_divisions; }
  int num_elevation_divisions() const { return l_.num_elevation_divisions; }
  int cols() const { return l_.num_azimuth_divisions; } int rows() const { return l_.num_elevation_divisions; }
  float min_valid_range_m() const { return l_.min_valid_range_m; }
  float vertical_fov_rad() const { return l_.max_elevation_rad - l_.min_elevation_rad; }
  const nvbx_lidar& c_abi() const { return l_; }
  bool operator==(const Lidar& o) const {
    return l_.num_azimuth_divisions == o.l_.num_azimuth_divisions && l_.num_elevation_divisions == o.l_.num_elevation_divisions &&
           l_.min_valid_range_m == o.l_.min_valid_range_m && l_.min_elevation_rad == o.l_.min_elevation_rad && l_.max_elevation_rad == o.l_.max_elevation_rad;
  }
 private:
  nvbx_lidar l_{0, 0, 0.f, 0.f, 0.f};
};

}  // namespace nvblox
