// nvblox/sensors/pointcloud.h -- device-resident point list (Vector3f) as handed to
// MultiMapper::integrateDepth(const Pointcloud&, const Transform&, const Lidar&, ...) at nvblox_node.cpp:1382-1384.
#pragma once
#include <hip/hip_runtime_api.h>
#include <vector>
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"
#include "nvblox/map/layer.h"
#include "nvblox/sensors/camera.h"
#include "nvblox/sensors/image.h"
#include "nvblox_hip.h"

namespace nvblox {

class Pointcloud {
 public:
  explicit Pointcloud(MemoryType memory_type = MemoryType::kDevice) : memory_type_(memory_type) {}
  ~Pointcloud() { if (data_) (void)hipFree(data_); if (times_) (void)hipFree(times_); }
  Pointcloud(const Pointcloud&) = delete;
  Pointcloud& operator=(const Pointcloud&) = delete;
  int size() const { return (int)size_; }
  bool empty() const { return size_ == 0; }
  const Vector3f* dataConstPtr() const { return data_; }
  Vector3f* dataPtr() { return data_; }
  MemoryType memory_type() const { return memory_type_; }
  void resizeAsync(size_t n, const CudaStream&) { resize(n); }
  void resize(size_t n) {
    if (n > cap_) { if (data_) (void)hipFree(data_); (void)hipMalloc((void**)&data_, n * sizeof(Vector3f)); cap_ = n; }
    size_ = n;
  }
  void copyFromAsync(const std::vector<Vector3f>& points, const CudaStream& stream) { copyFromAsync(points.data(), points.size(), stream); }
  // per-point time within the scan in milliseconds (loaded only for LiDAR motion compensation, nvblox_node.cpp:1339-1348)
  void copyTimestampsFromAsync(const float* rel_time_ms, size_t n, const CudaStream& stream) {
    if (n > times_cap_) { if (times_) (void)hipFree(times_); (void)hipMalloc((void**)&times_, n * sizeof(float)); times_cap_ = n; }
    if (n) (void)hipMemcpyAsync(times_, rel_time_ms, n * sizeof(float), hipMemcpyDefault, stream);
    times_size_ = n;
  }
  bool hasTimestamps() const { return times_size_ == size_ && size_ > 0; }
  const float* timestampsConstPtr() const { return times_; }
  void copyFromAsync(const Vector3f* points, size_t n, const CudaStream& stream) {
    resizeAsync(n, stream);
    if (n) (void)hipMemcpyAsync(data_, points, n * sizeof(Vector3f), hipMemcpyDefault, stream);
  }
 private:
  Vector3f* data_ = nullptr;
  size_t size_ = 0, cap_ = 0;
  float* times_ = nullptr; size_t times_size_ = 0, times_cap_ = 0;
  MemoryType memory_type_;
};

// DepthImageBackProjector::backProjectOnGPU(depth, camera, &pointcloud_C, max_back_projection_distance) --
// nvblox_node.cpp:1128-1130,1170; fuser_node.cpp:294-296.  Default-constructed like the reference's member.
class DepthImageBackProjector {
 public:
  DepthImageBackProjector() = default;
  void backProjectOnGPU(const DepthImage& image, const Camera& camera, Pointcloud* pointcloud_C, float max_back_projection_distance_m = 0.0f) const {
    nvbx_mapper* m = detail::requireContextMapper("DepthImageBackProjector");
    const nvbx_camera c = camera.c_abi();
    const int64_t cap = (int64_t)image.rows() * image.cols();
    pointcloud_C->resize((size_t)cap);
    int64_t n = 0;
    if (cap > 0)
      checkNvbx(nvbx_backproject_depth(m, image.dataConstPtr(), image.rows(), image.cols(), &c, max_back_projection_distance_m,
                                       reinterpret_cast<float*>(pointcloud_C->dataPtr()), cap, &n), "nvbx_backproject_depth");
    pointcloud_C->resize((size_t)n);
  }
};

// transformPointcloudOnGPU(T_L_C, pointcloud_C, &pointcloud_L) -- nvblox_node.cpp:1131, fuser_node.cpp:297
inline void transformPointcloudOnGPU(const Transform& T_out_in, const Pointcloud& pointcloud_in, Pointcloud* pointcloud_out) {
  nvbx_mapper* m = detail::requireContextMapper("transformPointcloudOnGPU");
  pointcloud_out->resize((size_t)pointcloud_in.size());
  if (pointcloud_in.size() == 0) return;
  float T[16]; T_out_in.toRowMajor(T);
  checkNvbx(nvbx_transform_pointcloud(m, T, reinterpret_cast<const float*>(pointcloud_in.dataConstPtr()), pointcloud_in.size(),
                                      reinterpret_cast<float*>(pointcloud_out->dataPtr())), "nvbx_transform_pointcloud");
}

}  // namespace nvblox
