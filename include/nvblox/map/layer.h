// nvblox/map/layer.h -- Layer<VoxelBlock<T>> views over the HBM-resident voxel-block hash of one mapper.
// The reference's layers own their blocks; here the mapper owns all pools (one slot id addresses the TSDF, colour and
// ESDF block of an Index3D) and a layer object is a typed view.  Accessors as used by the reference:
// tsdf_layer().block_size() / voxel_size() (layer_publishing.cpp:706,727-728), numAllocatedBlocks, getAllBlockIndices,
// getBlockAtIndex, allocateBlockAtIndex, callFunctionOnAllVoxels, getAABBOfAllocatedBlocks
// (test_esdf_and_gradient_conversions.cpp:85-92,114,118).  Blocks come back as HOST copies (the reference hands out
// device pointers); device-side consumers use the C-ABI gather kernels instead (nvbx_esdf_dense_grid, slicer).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <vector>
#include "nvblox/core/types.h"
#include "nvblox/map/voxels.h"
#include "nvblox_hip.h"

namespace nvblox {

// checkCudaErrors convention of the reference: print + exit(99) (nvblox_ros_common/src/check_cuda_errors.cpp:24-32)
inline void checkNvbx(int rc, const char* what) {
  if (rc < 0) { std::fprintf(stderr, "[nvblox_hip] %s failed (%d): %s\n", what, rc, nvbx_last_error()); std::exit(99); }
}

// Free functions and default-constructed helpers of the reference (DepthImageBackProjector, transformPointcloudOnGPU) own no
// mapper; they run on the stream of the most recently created Mapper of the process (the node has one pipeline).
namespace detail {
inline nvbx_mapper*& contextMapper() { static nvbx_mapper* m = nullptr; return m; }
inline nvbx_mapper* requireContextMapper(const char* who) {
  if (!contextMapper()) { std::fprintf(stderr, "[nvblox_hip] %s needs a live nvblox::Mapper (none has been created)\n", who); std::abort(); }
  return contextMapper();
}
}  // namespace detail

template <typename VoxelType, uint32_t kLayer>
class BlockLayerView {
 public:
  using BlockType = VoxelBlock<VoxelType>;
  using VoxelT = VoxelType;
  BlockLayerView() = default;
  BlockLayerView(nvbx_mapper* m, float voxel_size) : m_(m), voxel_size_(voxel_size) {}
  float voxel_size() const { return voxel_size_; }
  float block_size() const { return voxel_size_ * 8.0f; }
  MemoryType memory_type() const { return MemoryType::kDevice; }
  int numAllocatedBlocks() const { const int64_t n = nvbx_num_blocks(m_, kLayer); checkNvbx((int)(n < 0 ? n : 0), "nvbx_num_blocks"); return (int)n; }
  size_t size() const { return (size_t)numAllocatedBlocks(); }
  std::vector<Index3D> getAllBlockIndices() const {
    std::vector<Index3D> out((size_t)numAllocatedBlocks());
    if (!out.empty()) {
      const int64_t n = nvbx_block_indices(m_, kLayer, reinterpret_cast<nvbx_index3d*>(out.data()), (int64_t)out.size());
      checkNvbx((int)(n < 0 ? n : 0), "nvbx_block_indices");
      out.resize((size_t)std::min<int64_t>(n, (int64_t)out.size()));
    }
    return out;
  }
  bool isBlockAllocated(const Index3D& idx) const { return getBlockAtIndex(idx) != nullptr; }
  std::shared_ptr<BlockType> getBlockAtIndex(const Index3D& idx) const {
    auto b = std::make_shared<BlockType>();
    const int rc = nvbx_get_block(m_, kLayer, nvbx_index3d{idx.x(), idx.y(), idx.z()}, b.get());
    if (rc == NVBX_E_NOTFOUND) return nullptr;
    checkNvbx(rc, "nvbx_get_block");
    return b;
  }
  std::shared_ptr<BlockType> allocateBlockAtIndex(const Index3D& idx) {
    auto b = getBlockAtIndex(idx);
    if (b) return b;
    b = std::make_shared<BlockType>();
    setBlockAtIndex(idx, *b);
    return b;
  }
  void setBlockAtIndex(const Index3D& idx, const BlockType& block) {
    checkNvbx(nvbx_set_block(m_, kLayer, nvbx_index3d{idx.x(), idx.y(), idx.z()}, &block), "nvbx_set_block");
  }
  nvbx_mapper* c_handle() const { return m_; }
 private:
  nvbx_mapper* m_ = nullptr;
  float voxel_size_ = 0.f;
};

using TsdfLayer = BlockLayerView<TsdfVoxel, NVBX_LAYER_TSDF>;
using OccupancyLayer = BlockLayerView<OccupancyVoxel, NVBX_LAYER_OCCUPANCY>;
using FreespaceLayer = BlockLayerView<FreespaceVoxel, NVBX_LAYER_FREESPACE>;
using ColorLayer = BlockLayerView<ColorVoxel, NVBX_LAYER_COLOR>;
using EsdfLayer = BlockLayerView<EsdfVoxel, NVBX_LAYER_ESDF>;

// callFunctionOnAllVoxels<VoxelType>(layer*, fn(block_idx, voxel_idx, VoxelType*)) -- read-modify-write of every voxel
template <typename VoxelType, typename LayerT>
void callFunctionOnAllVoxels(LayerT* layer, std::function<void(const Index3D&, const Index3D&, VoxelType*)> fn) {
  for (const Index3D& bi : layer->getAllBlockIndices()) {
    auto b = layer->getBlockAtIndex(bi);
    if (!b) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) fn(bi, Index3D(x, y, z), &b->voxels[x][y][z]);
    layer->setBlockAtIndex(bi, *b);
  }
}
template <typename VoxelType, typename LayerT>
void callFunctionOnAllVoxels(const LayerT& layer, std::function<void(const Index3D&, const Index3D&, const VoxelType*)> fn) {
  for (const Index3D& bi : layer.getAllBlockIndices()) {
    auto b = layer.getBlockAtIndex(bi);
    if (!b) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) fn(bi, Index3D(x, y, z), &b->voxels[x][y][z]);
  }
}

template <typename LayerT>
AxisAlignedBoundingBox getAABBOfAllocatedBlocks(const LayerT& layer) {
  const auto idx = layer.getAllBlockIndices();
  if (idx.empty()) return AxisAlignedBoundingBox();
  Index3D mn = idx[0], mx = idx[0];
  for (const auto& i : idx) for (int a = 0; a < 3; a++) { if (i[a] < mn[a]) mn[a] = i[a]; if (i[a] > mx[a]) mx[a] = i[a]; }
  const float bs = layer.block_size();
  return AxisAlignedBoundingBox({mn.x() * bs, mn.y() * bs, mn.z() * bs}, {(mx.x() + 1) * bs, (mx.y() + 1) * bs, (mx.z() + 1) * bs});
}

}  // namespace nvblox
