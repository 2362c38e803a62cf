// nvblox/map/unified_3d_grid.h -- Unified3DGrid<T>: a dense grid over an AABB of GLOBAL voxel indices, x-major
// (linear = ((x - min.x) * Ny + (y - min.y)) * Nz + (z - min.z)), as nvblox_ros uses it:
//   esdf_and_gradients_conversions.hpp:41,74-75 (gpu_grid_(kDevice), cpu_grid_(kHost)), .cu:95-123 (copyFromAsync, aabb_size,
//   min_index, data().toVectorAsync), test_esdf_and_gradient_conversions.cpp:36-83 (setAABB, operator()(Index3D)).
// and voxelLayerToDenseVoxelGridInAABBAsync for the ESDF -> signed-distance conversion (.cu:88-93), which is one launch of
// libnvblox_hip.so (nvbx_esdf_dense_grid) instead of a device functor template.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <vector>
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"
#include "nvblox/map/layer.h"

namespace nvblox {

template <typename T>
class Unified3DGrid {
 public:
  // what `grid.data()` hands out: enough of unified_vector<T> for `.toVectorAsync(stream)` / `.data()`
  class DataView {
   public:
    DataView(const T* p, size_t n) : p_(p), n_(n) {}
    std::vector<T> toVectorAsync(const CudaStream& stream) const {
      std::vector<T> out(n_);
      if (n_) { (void)hipMemcpyAsync(out.data(), p_, n_ * sizeof(T), hipMemcpyDefault, stream); stream.synchronize(); }
      return out;
    }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
   private:
    const T* p_; size_t n_;
  };

  explicit Unified3DGrid(MemoryType memory_type = MemoryType::kUnified) : memory_type_(memory_type) {}
  ~Unified3DGrid() { release(); }
  Unified3DGrid(const Unified3DGrid&) = delete;
  Unified3DGrid& operator=(const Unified3DGrid&) = delete;

  // (re)allocates; contents are zero for host-visible memory types
  void setAABB(const Index3D& min_index, const Index3D& size) {
    min_ = min_index; size_ = size;
    const size_t need = numel();
    if (need > cap_) {
      release();
      if (memory_type_ == MemoryType::kHost) (void)hipHostMalloc((void**)&data_, need * sizeof(T));
      else if (memory_type_ == MemoryType::kUnified) (void)hipMallocManaged((void**)&data_, need * sizeof(T));
      else (void)hipMalloc((void**)&data_, need * sizeof(T));
      cap_ = need;
    }
    if (memory_type_ != MemoryType::kDevice) for (size_t i = 0; i < need; i++) data_[i] = T();
  }
  const Index3D& min_index() const { return min_; }
  const Index3D& aabb_size() const { return size_; }
  size_t numel() const { return (size_t)size_.x() * (size_t)size_.y() * (size_t)size_.z(); }
  MemoryType memory_type() const { return memory_type_; }
  bool isInGrid(const Index3D& idx) const {
    for (int a = 0; a < 3; a++) if (idx[a] < min_[a] || idx[a] >= min_[a] + size_[a]) return false;
    return true;
  }
  size_t linearIndex(const Index3D& idx) const {
    return ((size_t)(idx.x() - min_.x()) * size_.y() + (size_t)(idx.y() - min_.y())) * size_.z() + (size_t)(idx.z() - min_.z());
  }
  // host access (kUnified / kHost grids), global voxel index
  T& operator()(const Index3D& idx) { return data_[linearIndex(idx)]; }
  const T& operator()(const Index3D& idx) const { return data_[linearIndex(idx)]; }
  DataView data() const { return DataView(data_, numel()); }
  T* dataPtr() { return data_; }
  const T* dataConstPtr() const { return data_; }
  void copyFromAsync(const Unified3DGrid& other, const CudaStream& stream) {
    setAABB(other.min_, other.size_);
    if (numel()) (void)hipMemcpyAsync(data_, other.data_, numel() * sizeof(T), hipMemcpyDefault, stream);
  }

 private:
  void release() {
    if (!data_) return;
    if (memory_type_ == MemoryType::kHost) (void)hipHostFree(data_); else (void)hipFree(data_);
    data_ = nullptr; cap_ = 0;
  }
  MemoryType memory_type_;
  Index3D min_{0, 0, 0}, size_{0, 0, 0};
  T* data_ = nullptr;
  size_t cap_ = 0;
};

// What esdf_and_gradients_conversions.cu:33-48 expresses as a device functor: an observed ESDF voxel becomes
// +-sqrt(squared_distance_vox) * voxel_size (negative inside), anything else `default_value`.
struct SignedDistanceConversion { float voxel_size; float default_value; };

// voxelLayerToDenseVoxelGridInAABBAsync(esdf_layer, aabb, default_value, conversion_op, &gpu_grid, stream) -- .cu:88-93.
// The grid covers the voxels floor(aabb.min / voxel_size) .. floor(aabb.max / voxel_size) INCLUSIVE (the reference's test walks
// exactly that range, test_esdf_and_gradient_conversions.cpp:134-140).  Runs on the mapper's stream and is complete on return,
// so work the caller enqueues on `stream` afterwards (cpu_grid.copyFromAsync) sees the result.
inline void voxelLayerToDenseVoxelGridInAABBAsync(const EsdfLayer& esdf_layer, const AxisAlignedBoundingBox& aabb, float default_value,
                                                  const SignedDistanceConversion& /*conversion_op*/, Unified3DGrid<float>* grid_ptr,
                                                  const CudaStream& /*stream*/) {
  const float vs = esdf_layer.voxel_size();
  Index3D mn, sz;
  for (int a = 0; a < 3; a++) {
    mn[a] = (int)std::floor(aabb.min()[a] / vs);
    sz[a] = (int)std::floor(aabb.max()[a] / vs) - mn[a] + 1;
    if (sz[a] < 1) sz[a] = 1;
  }
  grid_ptr->setAABB(mn, sz);
  const int32_t min_vox[3] = {mn.x(), mn.y(), mn.z()}, size_vox[3] = {sz.x(), sz.y(), sz.z()};
  float* out = grid_ptr->dataPtr();
  Unified3DGrid<float> staging(MemoryType::kDevice);
  if (grid_ptr->memory_type() == MemoryType::kHost) { staging.setAABB(mn, sz); out = staging.dataPtr(); }     // the kernel writes device-visible memory
  checkNvbx(nvbx_esdf_dense_grid(esdf_layer.c_handle(), min_vox, size_vox, default_value, out), "nvbx_esdf_dense_grid");
  checkNvbx(nvbx_synchronize(esdf_layer.c_handle()), "nvbx_synchronize");
  if (out != grid_ptr->dataPtr()) (void)hipMemcpy(grid_ptr->dataPtr(), out, grid_ptr->numel() * sizeof(float), hipMemcpyDefault);
}

}  // namespace nvblox
