// nvblox/sensors/image.h -- Image<T> (DepthImage, ColorImage, MonoImage): row-major, device-resident by default, as
// used at nvblox_node.hpp:485-488 (node-owned reusable buffers), nvblox_node.cpp:835 `Image<float> img(MemoryType::kDevice)`,
// esdf_slice_conversions.cu:85-108 (.cols() .rows() .numel() .dataConstPtr()), image_conversions.cpp:147-155
// (.copyFromAsync(rows, cols, ptr, stream)).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <utility>
#include "nvblox_hip.h"
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"

namespace nvblox {

// Device images (MemoryType::kDevice, the default) live in library-owned, reference-counted frames (nvbx_frame_acquire, include/nvblox_hip.h).
// Why: a mapper with colour deferral on (the default) HOLDS an integrateColor back until the next integrateDepth.  The reference's node refills its one
// colour image right before every integrateColor (nvblox_node.cpp:1237-1263; the converter writes through the non-const dataPtr(),
// conversions/image_conversions_thrust.cu:75-80).  Here the mapper RETAINS the frame of a held-back image instead of copying it, and every WRITE access
// of the image -- non-const dataPtr(), copyFromAsync, resize -- first makes sure nobody else holds the frame: if a mapper does, the image lets go of it
// and continues in another frame of the pool (rotation; the mapper's frame stays untouched until its launches are done, then returns to the pool).
// What differs from a plain buffer: after integrateColor, the memory behind a NON-CONST dataPtr() is a different frame whose contents are unspecified
// (every writer in the reference overwrites the whole image); NVBX_IMAGE_ROTATE_PRESERVE=1 in the environment copies the old contents over first
// (a blocking device-to-device copy -- for hosts that patch images in place).  Const access (dataConstPtr) never rotates.
// Lifetime (round 6): an image that is destroyed, resized or rotated lets go of its frame with nvbx_frame_release -- the frame is handed to another
// image only after everything enqueued so far on the streams the library knows (every live mapper's stream -- the node's cuda_stream_ -- and the
// legacy default stream) has finished: a temporary DepthImage passed to integrateDepth may die right after the call, as with the reference's buffers
// (whose cudaFree waits for the device).  Work on a non-blocking stream of the host's own that no mapper runs on is the
// host's to finish first, as it is with any allocator that does not synchronise.  The frame stays on the device it was acquired on: a rotation or a
// regrowth happens there, whatever the calling thread's current device is.
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
    if (this != &o) { release(); data_ = o.data_; rows_ = o.rows_; cols_ = o.cols_; cap_ = o.cap_; memory_type_ = o.memory_type_; device_ = o.device_; o.data_ = nullptr; o.rows_ = o.cols_ = 0; o.cap_ = 0; }
    return *this;
  }
  int rows() const { return rows_; } int cols() const { return cols_; }
  int height() const { return rows_; } int width() const { return cols_; }
  int numel() const { return rows_ * cols_; }
  MemoryType memory_type() const { return memory_type_; }
  T* dataPtr() { makeExclusive(NVBX_STREAM_UNKNOWN); return data_; }
  const T* dataConstPtr() const { return data_; }
  // (re)allocate without preserving contents; keeps the allocation when it is large enough
  void resize(int rows, int cols) {
    const size_t need = (size_t)rows * cols;
    if (need > cap_) {
      release();
      if (memory_type_ == MemoryType::kHost) (void)hipHostMalloc((void**)&data_, need * sizeof(T));
      else if (memory_type_ == MemoryType::kUnified) (void)hipMallocManaged((void**)&data_, need * sizeof(T));
      else {
        int dev = device_; if (dev < 0) (void)hipGetDevice(&dev);      // (an image that had a frame regrows on that frame's device)
        void* p = nullptr; if (nvbx_frame_acquire(dev, need * sizeof(T), NVBX_STREAM_UNKNOWN, &p) == 0) { data_ = static_cast<T*>(p); device_ = dev; }
      }
      cap_ = data_ ? need : 0;
    }
    rows_ = rows; cols_ = cols;
  }
  void copyFromAsync(int rows, int cols, const T* src, const CudaStream& stream) {
    resize(rows, cols);
    makeExclusive(static_cast<void*>(static_cast<hipStream_t>(stream)));      // (the whole image is overwritten: nothing to preserve)
    (void)hipMemcpyAsync(data_, src, (size_t)rows * cols * sizeof(T), hipMemcpyDefault, stream);
  }
  void copyToAsync(T* dst, const CudaStream& stream) const {
    (void)hipMemcpyAsync(dst, data_, (size_t)numel() * sizeof(T), hipMemcpyDefault, stream);
  }
  // libnvblox_hip extension: the frame is shared with a mapper that holds the image back (the next write access rotates)
  bool sharedWithMapper() const { return memory_type_ == MemoryType::kDevice && data_ && nvbx_frame_refcount(data_) > 1; }
 private:
  // before a write: if a mapper has retained the frame (a held-back integrateColor), or has let go of it but its launches may still be reading it and
  // were not enqueued on the writer's own stream, continue in another one
  void makeExclusive(void* writer_stream) {
    if (memory_type_ != MemoryType::kDevice || !data_ || nvbx_frame_writable(data_, writer_stream) != 0) return;
    int dev = nvbx_frame_device(data_); if (dev < 0) (void)hipGetDevice(&dev);      // (the frame's own device, not the calling thread's current one)
    void* p = nullptr;
    if (nvbx_frame_acquire(dev, cap_ * sizeof(T), writer_stream, &p) != 0) return;       // (no memory: stay -- the mapper's staged copy is not in play, so say so loudly)
    static const bool preserve = [] { const char* e = getenv("NVBX_IMAGE_ROTATE_PRESERVE"); return e && e[0] == '1'; }();
    if (preserve && writer_stream == NVBX_STREAM_UNKNOWN) (void)hipMemcpy(p, data_, cap_ * sizeof(T), hipMemcpyDeviceToDevice);
    (void)nvbx_frame_release(data_);
    data_ = static_cast<T*>(p);
  }
  void release() {
    if (!data_) return;
    if (memory_type_ == MemoryType::kHost) (void)hipHostFree(data_);
    else if (memory_type_ == MemoryType::kUnified) (void)hipFree(data_);
    else (void)nvbx_frame_release(data_);
    data_ = nullptr; cap_ = 0;
  }
  T* data_ = nullptr;
  int rows_ = 0, cols_ = 0;
  size_t cap_ = 0;
  int device_ = -1;          // the device of the image's frame (kDevice), remembered across release() so that a regrowth stays there
  MemoryType memory_type_;
};

using DepthImage = Image<float>;
using ColorImage = Image<Color>;
using MonoImage = Image<uint8_t>;

}  // namespace nvblox
