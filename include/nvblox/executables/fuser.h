// nvblox/executables/fuser.h -- the offline fuser FuserNode drives (nvblox_ros/include/nvblox_ros/fuser_node.hpp:41,94;
// src/lib/fuser_node.cpp:48-65 createFuser, :85-97 setMultiMapper / multi_mapper() / static_mapper(), :216 integrateFrame,
// :229-232 getSensorData / getColorFrame / getSensor / getSensorPose, :281 getSerializedColorMesh; rosbag_data_loader.cpp:101
// `std::make_unique<CameraFuser>(std::move(data_loader), kDontInitializeFromGflagsFlag)`).
// [U] Fuser<SensorType>::integrateFrame restated: load the next frame from the data loader; integrate depth every
// tsdf_frame_subsampling-th frame, colour every color_frame_subsampling-th, update the mesh every mesh_frame_subsampling-th and the
// ESDF every esdf_frame_subsampling-th (0 = never); timers carry the reference's tags.  No gflags here: `init_from_gflags` is
// accepted and ignored (the node always passes false).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include "nvblox/datasets/data_loader_interface.h"
#include "nvblox/mapper/multi_mapper.h"
#include "nvblox/mesh/mesh.h"
#include "nvblox/utils/timing.h"

namespace nvblox {

template <typename SensorType>
class Fuser {
 public:
  Fuser(std::unique_ptr<datasets::RgbdDataLoaderInterface>&& data_loader, bool init_from_gflags = false) : data_loader_(std::move(data_loader)) {
    (void)init_from_gflags;
    depth_frame_ = std::make_shared<DepthImage>(MemoryType::kDevice); color_frame_ = std::make_shared<ColorImage>(MemoryType::kDevice);
    sensor_ = std::make_shared<SensorType>(); T_L_C_ = std::make_shared<Transform>();
  }
  void setMultiMapper(std::shared_ptr<MultiMapper> multi_mapper) { multi_mapper_ = std::move(multi_mapper); }      // fuser_node.cpp:85
  std::shared_ptr<MultiMapper> multi_mapper() const { return multi_mapper_; }                                     // :94
  std::shared_ptr<Mapper> static_mapper() const { return multi_mapper_ ? multi_mapper_->background_mapper() : nullptr; }   // :97

  // frame subsampling ([U] the core's fuser flags of the same names)
  int tsdf_frame_subsampling = 1, color_frame_subsampling = 1, mesh_frame_subsampling = 1, esdf_frame_subsampling = 1;

  // fuser_node.cpp:216: kBadFrame = skip and keep going, kNoMoreData = finished
  datasets::DataLoadResult integrateFrame(const int frame_number) {
    if (!data_loader_ || !multi_mapper_) return datasets::DataLoadResult::kNoMoreData;
    timing::Timer t_file("fuser/file_loading");
    Transform T_L_color; SensorType color_sensor;
    const datasets::DataLoadResult r = data_loader_->loadNext(depth_frame_.get(), T_L_C_.get(), sensor_.get(), color_frame_.get(), &T_L_color, &color_sensor,
                                                             nullptr, nullptr, nullptr);
    t_file.Stop();
    if (r != datasets::DataLoadResult::kSuccess) return r;
    timing::Timer t("fuser/integrate_frame");
    if (tsdf_frame_subsampling > 0 && frame_number % tsdf_frame_subsampling == 0) multi_mapper_->integrateDepth(*depth_frame_, *T_L_C_, *sensor_);
    if (color_frame_subsampling > 0 && frame_number % color_frame_subsampling == 0 && color_frame_->numel() > 0)
      multi_mapper_->integrateColor(*color_frame_, T_L_color, color_sensor);
    if (mesh_frame_subsampling > 0 && frame_number % mesh_frame_subsampling == 0) multi_mapper_->updateColorMesh();
    if (esdf_frame_subsampling > 0 && frame_number % esdf_frame_subsampling == 0) multi_mapper_->updateEsdf();
    return r;
  }
  // [U] Fuser::run(): every frame until the loader is exhausted; returns the number of frames integrated
  int integrateFrames(int max_frames = -1) {
    int n = 0;
    for (int i = 0; max_frames < 0 || i < max_frames; i++) {
      const datasets::DataLoadResult r = integrateFrame(i);
      if (r == datasets::DataLoadResult::kNoMoreData) break;
      if (r == datasets::DataLoadResult::kSuccess) n++;
    }
    return n;
  }
  // what FuserNode publishes after a step (fuser_node.cpp:229-232)
  std::shared_ptr<const DepthImage> getSensorData() const { return depth_frame_; }
  std::shared_ptr<const ColorImage> getColorFrame() const { return color_frame_; }
  std::shared_ptr<const SensorType> getSensor() const { return sensor_; }
  std::shared_ptr<const Transform> getSensorPose() const { return T_L_C_; }
  // fuser_node.cpp:280-281: the mesh blocks of the last mesh update, serialized for conversions::meshMessageFromSerializedMesh
  std::shared_ptr<SerializedColorMeshLayer> getSerializedColorMesh() {
    std::shared_ptr<Mapper> m = static_mapper();
    m->serializeSelectedLayers(LayerType::kColorMesh, -1.0f);
    return m->serializedColorMeshLayer();
  }

 private:
  std::unique_ptr<datasets::RgbdDataLoaderInterface> data_loader_;
  std::shared_ptr<MultiMapper> multi_mapper_;
  std::shared_ptr<DepthImage> depth_frame_; std::shared_ptr<ColorImage> color_frame_;
  std::shared_ptr<SensorType> sensor_; std::shared_ptr<Transform> T_L_C_;
};
using CameraFuser = Fuser<Camera>;      // fuser_node.hpp:94

}  // namespace nvblox
