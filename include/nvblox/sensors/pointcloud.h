// nvblox/sensors/pointcloud.h -- device-resident point list (Vector3f) as handed to
// MultiMapper::integrateDepth(const Pointcloud&, const Transform&, const Lidar&, ...) at nvblox_node.cpp:1382-1384.
#pragma once
#include <hip/hip_runtime_api.h>
#include <vector>
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"

namespace nvblox {

class Pointcloud {
 public:
  explicit Pointcloud(MemoryType memory_type = MemoryType::kDevice) : memory_type_(memory_type) {}
  ~Pointcloud() { if (data_) (void)hipFree(data_); }
  Pointcloud(const Pointcloud&) = delete;
  Pointcloud& operator=(const Pointcloud&) = delete;
  int size() const { return (int)size_; }
  bool empty() const { return size_ == 0; }
  const Vector3f* dataConstPtr() const { return data_; }
  Vector3f* dataPtr() { return data_; }
  MemoryType memory_type() const { return memory_type_; }
  void resizeAsync(size_t n, const CudaStream&) {
    if (n > cap_) { if (data_) (void)hipFree(data_); (void)hipMalloc((void**)&data_, n * sizeof(Vector3f)); cap_ = n; }
    size_ = n;
  }
  void copyFromAsync(const std::vector<Vector3f>& points, const CudaStream& stream) { copyFromAsync(points.data(), points.size(), stream); }
  void copyFromAsync(const Vector3f* points, size_t n, const CudaStream& stream) {
    resizeAsync(n, stream);
    if (n) (void)hipMemcpyAsync(data_, points, n * sizeof(Vector3f), hipMemcpyDefault, stream);
  }
 private:
  Vector3f* data_ = nullptr;
  size_t size_ = 0, cap_ = 0;
  MemoryType memory_type_;
};

}  // namespace nvblox
