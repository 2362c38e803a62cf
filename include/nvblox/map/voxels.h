// nvblox/map/voxels.h -- voxel structs with the field names the reference's consumers read
// (layer_publishing.cpp:62-76,111,179,192; conversions/esdf_and_gradients_conversions.cu:33-44;
// test_esdf_and_gradient_conversions.cpp:89-90).
#pragma once
#include "nvblox/core/types.h"
#include "nvblox_hip.h"

namespace nvblox {

struct TsdfVoxel { float distance = 0.f; float weight = 0.f; };
struct OccupancyVoxel { float log_odds = 0.f; };      // layer_publishing.cpp:140-154
// layer_publishing.cpp:129-137,158-165
struct FreespaceVoxel { Time last_occupied_timestamp_ms; Time consecutive_occupancy_duration_ms; bool is_high_confidence_freespace = false; uint8_t initialized_ = 0; uint8_t pad_[6] = {0, 0, 0, 0, 0, 0}; };
static_assert(sizeof(FreespaceVoxel) == sizeof(nvbx_freespace_voxel), "C-ABI block copies are memcpy");
struct ColorVoxel { Color color; uint8_t pad_ = 0; float weight = 0.f; };
struct EsdfVoxel {
  float squared_distance_vox = 0.f;
  Vector3i parent_direction;
  bool is_inside = false, observed = false, is_site = false;
  uint8_t pad_ = 0;
};
static_assert(sizeof(TsdfVoxel) == sizeof(nvbx_tsdf_voxel), "C-ABI block copies are memcpy");
static_assert(sizeof(ColorVoxel) == sizeof(nvbx_color_voxel), "C-ABI block copies are memcpy");
static_assert(sizeof(EsdfVoxel) == sizeof(nvbx_esdf_voxel), "C-ABI block copies are memcpy");

// 8 x 8 x 8 voxels, indexed [x][y][z]  (linear z + 8 y + 64 x, layer_publishing.cpp:335,501)
template <typename VoxelType>
struct VoxelBlock {
  static constexpr int kVoxelsPerSide = 8;
  static constexpr int kNumVoxels = 512;
  VoxelType voxels[8][8][8];
};
using TsdfBlock = VoxelBlock<TsdfVoxel>;
using ColorBlock = VoxelBlock<ColorVoxel>;
using EsdfBlock = VoxelBlock<EsdfVoxel>;

}  // namespace nvblox
