// nvblox/core/cuda_stream.h -- the reference's CudaStream family over hipStream_t (call sites: nvblox_node.cpp:91
// CudaStream::createCudaStream(type); esdf_slice_conversions.cu:76-78,107-108 CudaStreamOwning, implicit conversion to
// the raw stream, .synchronize()).  The type names are kept so nvblox_ros compiles unchanged; the handle is a HIP stream.
#pragma once
#include <hip/hip_runtime_api.h>
#include <memory>

namespace nvblox {

// node_params.hpp:37-41: stream type 0..3
enum class CudaStreamType { kLegacyDefault = 0, kBlocking = 1, kNonBlocking = 2, kPerThread = 3 };

class CudaStream {
 public:
  virtual ~CudaStream() = default;
  hipStream_t& get() { return stream_; }
  operator hipStream_t() const { return stream_; }
  void synchronize() const { (void)hipStreamSynchronize(stream_); }
  static inline std::shared_ptr<CudaStream> createCudaStream(CudaStreamType type);
 protected:
  CudaStream() = default;
  hipStream_t stream_ = nullptr;
};

class CudaStreamOwning : public CudaStream {
 public:
  explicit CudaStreamOwning(unsigned int flags = hipStreamNonBlocking) { (void)hipStreamCreateWithFlags(&stream_, flags); }
  ~CudaStreamOwning() override { if (stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); } }
  CudaStreamOwning(const CudaStreamOwning&) = delete;
  CudaStreamOwning& operator=(const CudaStreamOwning&) = delete;
};

class CudaStreamNonOwning : public CudaStream {
 public:
  explicit CudaStreamNonOwning(hipStream_t s) { stream_ = s; }
};

inline std::shared_ptr<CudaStream> CudaStream::createCudaStream(CudaStreamType type) {
  switch (type) {
    case CudaStreamType::kLegacyDefault: return std::make_shared<CudaStreamNonOwning>(nullptr);
    case CudaStreamType::kPerThread: return std::make_shared<CudaStreamNonOwning>(hipStreamPerThread);
    case CudaStreamType::kBlocking: return std::make_shared<CudaStreamOwning>(hipStreamDefault);
    case CudaStreamType::kNonBlocking: default: return std::make_shared<CudaStreamOwning>(hipStreamNonBlocking);
  }
}

inline void warmupCuda() { (void)hipFree(nullptr); }   // fuser_node_main.cpp:38

}  // namespace nvblox
