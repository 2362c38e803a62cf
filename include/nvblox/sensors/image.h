// nvblox/sensors/image.h -- Image<T> (DepthImage, ColorImage, MonoImage): row-major, device-resident by default, as
// used at nvblox_node.hpp:485-488 (node-owned reusable buffers), nvblox_node.cpp:835 `Image<float> img(MemoryType::kDevice)`,
// esdf_slice_conversions.cu:85-108 (.cols() .rows() .numel() .dataConstPtr()), image_conversions.cpp:147-155
// (.copyFromAsync(rows, cols, ptr, stream)).
#pragma once
#include <hip/hip_runtime_api.h>
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"

namespace nvblox {

template <typename T>
class Image {
 public:
  using ElementType = T;
  explicit Image(MemoryType memory_type = MemoryType::kDevice) : memory_type_(memory_type) {}
  Image(int rows, int cols, MemoryType memory_type = MemoryType::kDevice) : memory_type_(memory_type) { resize(rows, cols); }
  ~Image() { release(); }
  Image(const Image&) = delete;
  Image& operator=(const Image&) = delete;
  Image(Image&& o) noexcept { *this = std::move(o); }
  Image& operator=(Image&& o) noexcept {
    if (this != &o) { release(); data_ = o.data_; rows_ = o.rows_; cols_ = o.cols_; cap_ = o.cap_; memory_type_ = o.memory_type_; o.data_ = nullptr; o.rows_ = o.cols_ = 0; o.cap_ = 0; }
    return *this;
  }
  int rows() const { return rows_; } int cols() const { return cols_; }
  int height() const { return rows_; } int width() const { return cols_; }
  int numel() const { return rows_ * cols_; }
  MemoryType memory_type() const { return memory_type_; }
  T* dataPtr() { return data_; }
  const T* dataConstPtr() const { return data_; }
  // (re)allocate without preserving contents; keeps the allocation when it is large enough
  void resize(int rows, int cols) {
    const size_t need = (size_t)rows * cols;
    if (need > cap_) {
      release();
      if (memory_type_ == MemoryType::kHost) (void)hipHostMalloc((void**)&data_, need * sizeof(T));
      else if (memory_type_ == MemoryType::kUnified) (void)hipMallocManaged((void**)&data_, need * sizeof(T));
      else (void)hipMalloc((void**)&data_, need * sizeof(T));
      cap_ = need;
    }
    rows_ = rows; cols_ = cols;
  }
  void copyFromAsync(int rows, int cols, const T* src, const CudaStream& stream) {
    resize(rows, cols);
    (void)hipMemcpyAsync(data_, src, (size_t)rows * cols * sizeof(T), hipMemcpyDefault, stream);
  }
  void copyToAsync(T* dst, const CudaStream& stream) const {
    (void)hipMemcpyAsync(dst, data_, (size_t)numel() * sizeof(T), hipMemcpyDefault, stream);
  }
 private:
  void release() {
    if (!data_) return;
    if (memory_type_ == MemoryType::kHost) (void)hipHostFree(data_); else (void)hipFree(data_);
    data_ = nullptr; cap_ = 0;
  }
  T* data_ = nullptr;
  int rows_ = 0, cols_ = 0;
  size_t cap_ = 0;
  MemoryType memory_type_;
};

using DepthImage = Image<float>;
using ColorImage = Image<Color>;
using MonoImage = Image<uint8_t>;

}  // namespace nvblox
