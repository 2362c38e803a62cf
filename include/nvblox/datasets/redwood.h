// nvblox/datasets/redwood.h -- datasets::redwood::createFuser(base_path, init_from_gflags) (fuser_node.cpp:53; fuser_node.hpp:33).
// [U] The Redwood indoor RGB-D layout the core's loader reads: <base>/depth/%05d.png (16-bit, millimetres; 1-based numbering),
// <base>/image/%05d.jpg, <base>/pose_*/*.log trajectory: per frame one metadata line (three integers) followed by a 4x4 row-major
// camera-to-world matrix; intrinsics = the PrimeSense default fu = fv = 525, cu = 319.5, cv = 239.5 at 640x480.
// Colour: image/%05d.jpg (baseline JPEG, datasets/jpeg_decoder.h), or a .png / .ppm of the same stem; else depth-only.
#pragma once
#include <dirent.h>
#include <cmath>
#include <fstream>
#include <memory>
#include <string>
#include <vector>
#include "nvblox/datasets/replica.h"

namespace nvblox {
namespace datasets {
namespace redwood {

class DataLoader : public RgbdDataLoaderInterface {
 public:
  explicit DataLoader(const std::string& base_path, std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>())
      : base_path_(base_path), cuda_stream_(std::move(cuda_stream)) {
    // the trajectory: first *.log under a pose_* directory of the sequence (or <base>/trajectory.log)
    std::string log = base_path + "/trajectory.log";
    if (!internal::fileExists(log)) {
      log.clear();
      if (DIR* d = opendir(base_path.c_str())) {
        while (dirent* e = readdir(d)) {
          const std::string name = e->d_name;
          if (name.rfind("pose_", 0) != 0) continue;
          if (DIR* d2 = opendir((base_path + "/" + name).c_str())) {
            while (dirent* e2 = readdir(d2)) { const std::string n2 = e2->d_name; if (n2.size() > 4 && n2.substr(n2.size() - 4) == ".log") { log = base_path + "/" + name + "/" + n2; break; } }
            closedir(d2);
          }
          if (!log.empty()) break;
        }
        closedir(d);
      }
    }
    std::ifstream f(log);
    if (log.empty() || !f) { setup_success_ = false; return; }
    int a, b, c; float m[16];
    while (f >> a >> b >> c) {
      bool ok = true;
      for (int i = 0; i < 16; i++) if (!(f >> m[i])) { ok = false; break; }
      if (!ok) break;
      poses_.push_back(Transform::fromRowMajor(m));
    }
    if (poses_.empty()) setup_success_ = false;
  }
  static std::unique_ptr<DataLoader> create(const std::string& base_path, std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>()) {
    auto l = std::make_unique<DataLoader>(base_path, std::move(cuda_stream));
    if (!l->setup_success_) return nullptr;
    return l;
  }
  using RgbdDataLoaderInterface::loadNext;
  DataLoadResult loadNext(DepthImage* depth_frame_ptr, Transform* T_L_D_ptr, Camera* depth_camera_ptr, ColorImage* color_frame_ptr, Transform* T_L_C_ptr,
                          Camera* color_camera_ptr, Time*, Transform*, Time*) override {
    if (frame_ >= (int)poses_.size()) return DataLoadResult::kNoMoreData;
    const int i = frame_++;
    const std::string depth_path = base_path_ + "/depth/" + internal::numbered("%05d.png", i + 1);
    if (!internal::fileExists(depth_path)) return DataLoadResult::kNoMoreData;
    if (!load16BitDepthImage(depth_path, depth_frame_ptr, 1.0f / 1000.0f, *cuda_stream_, &depth_scratch_)) return DataLoadResult::kBadFrame;
    *T_L_D_ptr = poses_[(size_t)i];
    *depth_camera_ptr = Camera(525.0f, 525.0f, 319.5f, 239.5f, depth_frame_ptr->cols(), depth_frame_ptr->rows());
    if (T_L_C_ptr) *T_L_C_ptr = *T_L_D_ptr;
    if (color_camera_ptr) *color_camera_ptr = *depth_camera_ptr;
    if (color_frame_ptr && !internal::loadColorOfStem(base_path_ + "/image/" + internal::numbered("%05d", i + 1), color_frame_ptr, *cuda_stream_, &color_scratch_))
      color_frame_ptr->resize(0, 0);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) if (!std::isfinite((*T_L_D_ptr)(r, c))) return DataLoadResult::kBadFrame;
    return DataLoadResult::kSuccess;
  }

 private:
  std::string base_path_;
  std::shared_ptr<CudaStream> cuda_stream_;
  std::vector<Transform> poses_;
  int frame_ = 0;
  std::vector<float> depth_scratch_; std::vector<Color> color_scratch_;
};

inline std::unique_ptr<CameraFuser> createFuser(const std::string& base_path, bool init_from_gflags = false,
                                                std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>()) {
  auto loader = DataLoader::create(base_path, std::move(cuda_stream));
  if (!loader) return std::unique_ptr<CameraFuser>();
  return std::make_unique<CameraFuser>(std::move(loader), init_from_gflags);
}

}  // namespace redwood
}  // namespace datasets
}  // namespace nvblox
