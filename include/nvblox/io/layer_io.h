// nvblox/io/layer_io.h -- io::outputVoxelLayerToPly(layer, path) -> bool: the voxel layers as PLY point clouds with an
// intensity, as the save_ply service writes them into a folder (nvblox_node.cpp:1615-1628: tsdf, esdf, freespace).
// [U] (the writer lives in the absent core) one point per voxel that carries information, at the voxel centre:
//   TSDF       weight > 0                      intensity = distance [m]
//   ESDF       observed                        intensity = signed distance [m] (+-sqrt(squared_distance_vox) * voxel_size)
//   occupancy  log-odds != 0                   intensity = log-odds
//   freespace  high-confidence freespace       intensity = 1
// Any layer type with voxel_size() / block_size() / getAllBlockIndices() / getBlockAtIndex() works (the façade's layer views do).
#pragma once
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>
#include "nvblox/core/types.h"
#include "nvblox/map/voxels.h"

namespace nvblox {
namespace io {

namespace detail {
inline bool voxelIntensity(const TsdfVoxel& v, float, float* out) { if (!(v.weight > 0.f)) return false; *out = v.distance; return true; }
inline bool voxelIntensity(const EsdfVoxel& v, float voxel_size, float* out) {
  if (!v.observed) return false;
  const float d = std::sqrt(v.squared_distance_vox) * voxel_size; *out = v.is_inside ? -d : d; return true;
}
inline bool voxelIntensity(const OccupancyVoxel& v, float, float* out) { if (v.log_odds == 0.f) return false; *out = v.log_odds; return true; }
inline bool voxelIntensity(const FreespaceVoxel& v, float, float* out) { if (!v.is_high_confidence_freespace) return false; *out = 1.f; return true; }
}  // namespace detail

template <typename LayerT>
bool outputVoxelLayerToPly(const LayerT& layer, const std::string& filename) {
  struct P { float x, y, z, i; };
  std::vector<P> pts;
  const float vs = layer.voxel_size(), bs = layer.block_size();
  for (const Index3D& b : layer.getAllBlockIndices()) {
    const auto block = layer.getBlockAtIndex(b);
    if (!block) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      float intensity = 0.f;
      if (!detail::voxelIntensity(block->voxels[x][y][z], vs, &intensity)) continue;
      const Vector3f c = getCenterPositionFromBlockIndexAndVoxelIndex(bs, b, Index3D(x, y, z));
      pts.push_back({c.x(), c.y(), c.z(), intensity});
    }
  }
  FILE* f = std::fopen(filename.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nproperty float intensity\nend_header\n", pts.size());
  for (const P& p : pts) std::fprintf(f, "%.6f %.6f %.6f %.6f\n", p.x, p.y, p.z, p.i);
  return std::fclose(f) == 0;
}

}  // namespace io
}  // namespace nvblox
