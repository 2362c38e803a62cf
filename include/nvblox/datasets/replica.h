// nvblox/datasets/replica.h -- datasets::replica::createFuser(base_path, init_from_gflags) (fuser_node.cpp:56; fuser_node.hpp:34).
// [U] The Replica release the core's loader reads (the NICE-SLAM rendering of Replica): <base>/traj.txt = one row-major 4x4
// camera-to-world matrix per line; <base>/results/depth%06d.png = 16-bit depth, metres = raw / scale; <base>/results/frame%06d.jpg
// = colour; <base>/../cam_params.json (or <base>/cam_params.json) = {"camera": {"w","h","fx","fy","cx","cy","scale"}}.
// Colour: frame%06d.jpg (baseline JPEG, datasets/jpeg_decoder.h), or a .png / .ppm of the same stem; a frame whose colour does not decode
// (a progressive JPEG) is integrated depth-only.
#pragma once
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include "nvblox/datasets/data_loader_interface.h"
#include "nvblox/datasets/image_loader.h"
#include "nvblox/executables/fuser.h"

namespace nvblox {
namespace datasets {
namespace internal {
inline bool fileExists(const std::string& p) { FILE* f = std::fopen(p.c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; }
inline std::string numbered(const char* fmt, int i) { char b[64]; std::snprintf(b, sizeof(b), fmt, i); return b; }
// the value of "key": <number> in a flat JSON text
inline bool jsonNumber(const std::string& text, const std::string& key, double* out) {
  const size_t k = text.find("\"" + key + "\"");
  if (k == std::string::npos) return false;
  const size_t c = text.find(':', k);
  if (c == std::string::npos) return false;
  return std::sscanf(text.c_str() + c + 1, " %lf", out) == 1;
}
// colour frame of stem `stem` (no extension): JPEG (what the datasets ship), PNG, then PPM; false if none exists or decodes
inline bool loadColorOfStem(const std::string& stem, ColorImage* color, const CudaStream& stream, std::vector<Color>* scratch) {
  for (const char* ext : {".jpg", ".jpeg", ".png", ".ppm"})
    if (fileExists(stem + ext) && load8BitColorImage(stem + ext, color, stream, scratch)) return true;
  return false;
}
}  // namespace internal

namespace replica {

class DataLoader : public RgbdDataLoaderInterface {
 public:
  explicit DataLoader(const std::string& base_path, std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>())
      : base_path_(base_path), cuda_stream_(std::move(cuda_stream)) {
    std::string js;
    for (const std::string& p : {base_path + "/cam_params.json", base_path + "/../cam_params.json"}) {
      std::ifstream f(p);
      if (f) { std::stringstream ss; ss << f.rdbuf(); js = ss.str(); break; }
    }
    double w, h, fx, fy, cx, cy, scale;
    if (js.empty() || !internal::jsonNumber(js, "w", &w) || !internal::jsonNumber(js, "h", &h) || !internal::jsonNumber(js, "fx", &fx) ||
        !internal::jsonNumber(js, "fy", &fy) || !internal::jsonNumber(js, "cx", &cx) || !internal::jsonNumber(js, "cy", &cy) || !internal::jsonNumber(js, "scale", &scale)) {
      setup_success_ = false; return;
    }
    camera_ = Camera((float)fx, (float)fy, (float)cx, (float)cy, (int)w, (int)h); depth_scale_ = 1.0f / (float)scale;
    std::ifstream traj(base_path + "/traj.txt");
    if (!traj) { setup_success_ = false; return; }
    float m[16];
    while (true) {
      bool ok = true;
      for (int i = 0; i < 16; i++) if (!(traj >> m[i])) { ok = false; break; }
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
    const std::string depth_path = base_path_ + "/results/" + internal::numbered("depth%06d.png", i);
    if (!internal::fileExists(depth_path)) return DataLoadResult::kNoMoreData;
    if (!load16BitDepthImage(depth_path, depth_frame_ptr, depth_scale_, *cuda_stream_, &depth_scratch_)) return DataLoadResult::kBadFrame;
    *T_L_D_ptr = poses_[(size_t)i]; *depth_camera_ptr = camera_;
    if (T_L_C_ptr) *T_L_C_ptr = poses_[(size_t)i];
    if (color_camera_ptr) *color_camera_ptr = camera_;
    if (color_frame_ptr && !internal::loadColorOfStem(base_path_ + "/results/" + internal::numbered("frame%06d", i), color_frame_ptr, *cuda_stream_, &color_scratch_))
      color_frame_ptr->resize(0, 0);                       // no colour image that decodes (e.g. a progressive JPEG): depth-only frame
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) if (!std::isfinite((*T_L_D_ptr)(r, c))) return DataLoadResult::kBadFrame;
    return DataLoadResult::kSuccess;
  }

 private:
  std::string base_path_;
  std::shared_ptr<CudaStream> cuda_stream_;
  Camera camera_; float depth_scale_ = 1.0f / 6553.5f;
  std::vector<Transform> poses_;
  int frame_ = 0;
  std::vector<float> depth_scratch_; std::vector<Color> color_scratch_;
};

// fuser_node.cpp:56.  nullptr if the directory is not a Replica sequence.
inline std::unique_ptr<CameraFuser> createFuser(const std::string& base_path, bool init_from_gflags = false,
                                                std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>()) {
  auto loader = DataLoader::create(base_path, std::move(cuda_stream));
  if (!loader) return std::unique_ptr<CameraFuser>();
  return std::make_unique<CameraFuser>(std::move(loader), init_from_gflags);
}

}  // namespace replica
}  // namespace datasets
}  // namespace nvblox
