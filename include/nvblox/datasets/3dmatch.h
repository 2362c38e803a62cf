// nvblox/datasets/3dmatch.h -- datasets::threedmatch::createFuser(base_path, seq_id, init_from_gflags) (fuser_node.cpp:50;
// fuser_node.hpp:32).  [U] The 3DMatch / Sun3D layout the core's loader reads: <base>/camera-intrinsics.txt (3x3 K, row-major text),
// <base>/seq-%02d/frame-%06d.depth.png (16-bit, millimetres; 65535 = invalid), frame-%06d.color.png, frame-%06d.pose.txt (4x4
// camera-to-world).  All PNG: depth AND colour are decoded (image_loader.h).
#pragma once
#include <cmath>
#include <fstream>
#include <memory>
#include <string>
#include <vector>
#include "nvblox/datasets/replica.h"

namespace nvblox {
namespace datasets {
namespace threedmatch {

class DataLoader : public RgbdDataLoaderInterface {
 public:
  DataLoader(const std::string& base_path, int seq_id, std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>())
      : seq_dir_(base_path + "/" + internal::numbered("seq-%02d", seq_id)), cuda_stream_(std::move(cuda_stream)) {
    std::ifstream k(base_path + "/camera-intrinsics.txt");
    float m[9];
    for (int i = 0; i < 9; i++) if (!(k >> m[i])) { setup_success_ = false; return; }
    fu_ = m[0]; fv_ = m[4]; cu_ = m[2]; cv_ = m[5];
    if (!internal::fileExists(seq_dir_ + "/" + internal::numbered("frame-%06d.depth.png", 0))) setup_success_ = false;
  }
  static std::unique_ptr<DataLoader> create(const std::string& base_path, int seq_id, std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>()) {
    auto l = std::make_unique<DataLoader>(base_path, seq_id, std::move(cuda_stream));
    if (!l->setup_success_) return nullptr;
    return l;
  }
  using RgbdDataLoaderInterface::loadNext;
  DataLoadResult loadNext(DepthImage* depth_frame_ptr, Transform* T_L_D_ptr, Camera* depth_camera_ptr, ColorImage* color_frame_ptr, Transform* T_L_C_ptr,
                          Camera* color_camera_ptr, Time*, Transform*, Time*) override {
    const int i = frame_++;
    const std::string stem = seq_dir_ + "/" + internal::numbered("frame-%06d", i);
    if (!internal::fileExists(stem + ".depth.png")) return DataLoadResult::kNoMoreData;
    image_io::DecodedImage img;
    if (!image_io::decode(stem + ".depth.png", &img) || img.channels != 1) return DataLoadResult::kBadFrame;
    depth_scratch_.resize(img.data.size());
    for (size_t q = 0; q < img.data.size(); q++) depth_scratch_[q] = img.data[q] == 65535 ? 0.0f : (float)img.data[q] * (1.0f / 1000.0f);
    depth_frame_ptr->copyFromAsync(img.rows, img.cols, depth_scratch_.data(), *cuda_stream_);
    cuda_stream_->synchronize();
    std::ifstream p(stem + ".pose.txt");
    float m[16];
    for (int q = 0; q < 16; q++) if (!(p >> m[q]) || !std::isfinite(m[q])) return DataLoadResult::kBadFrame;      // (3DMatch marks lost tracking with inf / nan poses)
    *T_L_D_ptr = Transform::fromRowMajor(m);
    *depth_camera_ptr = Camera(fu_, fv_, cu_, cv_, img.cols, img.rows);
    if (T_L_C_ptr) *T_L_C_ptr = *T_L_D_ptr;
    if (color_camera_ptr) *color_camera_ptr = *depth_camera_ptr;
    if (color_frame_ptr && !(internal::fileExists(stem + ".color.png") && load8BitColorImage(stem + ".color.png", color_frame_ptr, *cuda_stream_, &color_scratch_)))
      color_frame_ptr->resize(0, 0);
    return DataLoadResult::kSuccess;
  }

 private:
  std::string seq_dir_;
  std::shared_ptr<CudaStream> cuda_stream_;
  float fu_ = 0.f, fv_ = 0.f, cu_ = 0.f, cv_ = 0.f;
  int frame_ = 0;
  std::vector<float> depth_scratch_; std::vector<Color> color_scratch_;
};

inline std::unique_ptr<CameraFuser> createFuser(const std::string& base_path, int seq_id, bool init_from_gflags = false,
                                                std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>()) {
  auto loader = DataLoader::create(base_path, seq_id, std::move(cuda_stream));
  if (!loader) return std::unique_ptr<CameraFuser>();
  return std::make_unique<CameraFuser>(std::move(loader), init_from_gflags);
}

}  // namespace threedmatch
}  // namespace datasets
}  // namespace nvblox
