// nvblox/mapper/multi_mapper.h -- nvblox::MultiMapper as constructed and driven by NvbloxNode / FuserNode
// (nvblox_node.cpp:187-210,781,1058-1062,1261-1264; fuser_node.cpp:85-94).  libnvblox_hip implements the static-TSDF
// mapping type (BASELINE.json north_star) and static occupancy (nvblox_base.yaml:9); the masked / dynamic / LiDAR-pointcloud overloads exist so the node
// compiles, and abort with a clear message if reached (the reference aborts on programmer errors).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <optional>
#include "nvblox/mapper/mapper.h"

namespace nvblox {

class MultiMapper {
 public:
  MultiMapper(float voxel_size_m, MappingType mapping_type, EsdfMode esdf_mode, MemoryType memory_type = MemoryType::kDevice,
              std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>(), int64_t block_capacity = Mapper::kDefaultBlockCapacity)
      : mapping_type_(mapping_type), esdf_mode_(esdf_mode), cuda_stream_(cuda_stream) {
    if (mapping_type != MappingType::kStaticTsdf && mapping_type != MappingType::kStaticOccupancy)
      unsupported("mapping types other than MappingType::kStaticTsdf / kStaticOccupancy");
    if (esdf_mode != EsdfMode::k2D) unsupported("EsdfMode::k3D");
    background_mapper_ = std::make_shared<Mapper>(voxel_size_m, memory_type,
                                                  mapping_type == MappingType::kStaticOccupancy ? ProjectiveLayerType::kOccupancy : ProjectiveLayerType::kTsdf,
                                                  cuda_stream, block_capacity);
    // the foreground (dynamic / human) mapper is never fed in static mode; keep a minimal one so the handle is valid
    foreground_mapper_ = std::make_shared<Mapper>(voxel_size_m, memory_type, ProjectiveLayerType::kNone, cuda_stream, 64);
  }
  void setMapperParams(const MapperParams& background, const MapperParams& foreground) { background_mapper_->setMapperParams(background); foreground_mapper_->setMapperParams(foreground); }
  void setMapperParams(const MapperParams& params) { background_mapper_->setMapperParams(params); }   // fuser_node.cpp:94
  void setMultiMapperParams(const MultiMapperParams& p) { multi_params_ = p; }
  std::shared_ptr<Mapper> background_mapper() const { return background_mapper_; }
  std::shared_ptr<Mapper> foreground_mapper() const { return foreground_mapper_; }

  void integrateDepth(const DepthImage& depth, const Transform& T_L_C, const Camera& camera, std::optional<Time> update_time_ms = std::nullopt) {
    (void)update_time_ms;   // consumed by the freespace layer only (dynamic mapping)
    background_mapper_->integrateDepth(depth, T_L_C, camera);
  }
  void integrateDepth(const DepthImage&, const MonoImage&, const Transform&, const Transform&, const Camera&, const Camera&) { unsupported("masked depth integration (human mapping)"); }
  // nvblox_node.cpp:1382-1384.  Per-point motion compensation (use_lidar_motion_compensation, node_params.hpp:152) needs
  // per-point timestamps that this Pointcloud does not carry: run the node with use_lidar_motion_compensation:=false.
  void integrateDepth(const Pointcloud& pointcloud, const Transform& T_L_C, const Lidar& lidar, bool use_lidar_motion_compensation = false,
                      std::optional<Transform> T_L_S_scan_end = std::nullopt, std::optional<Time> scan_duration_ms = std::nullopt,
                      std::optional<Time> update_time_ms = std::nullopt) {
    (void)T_L_S_scan_end; (void)scan_duration_ms; (void)update_time_ms;
    if (use_lidar_motion_compensation) unsupported("LiDAR motion compensation (set use_lidar_motion_compensation:=false)");
    background_mapper_->integrateLidarPointcloud(pointcloud, T_L_C, lidar);
  }
  const DepthImage& getLastDepthFrameFromPointcloud() const { return background_mapper_->getLastDepthFrameFromPointcloud(); }
  void integrateColor(const ColorImage& color, const Transform& T_L_C, const Camera& camera) { background_mapper_->integrateColor(color, T_L_C, camera); }
  void integrateColor(const ColorImage&, const MonoImage&, const Transform&, const Camera&) { unsupported("masked colour integration (human mapping)"); }
  void updateEsdf() { background_mapper_->updateEsdf(); }
  void updateColorMesh(UpdateFullLayer f = UpdateFullLayer::kNo) { background_mapper_->updateColorMesh(f); }

  MappingType mapping_type() const { return mapping_type_; }
  EsdfMode esdf_mode() const { return esdf_mode_; }

 private:
  [[noreturn]] static void unsupported(const char* what) {
    std::fprintf(stderr, "[nvblox_hip] %s is outside the MI355X hot path of this library (static TSDF + colour + 2-D ESDF + mesh)\n", what);
    std::abort();
  }
  MappingType mapping_type_; EsdfMode esdf_mode_;
  std::shared_ptr<CudaStream> cuda_stream_;
  std::shared_ptr<Mapper> background_mapper_, foreground_mapper_;
  MultiMapperParams multi_params_;
};

}  // namespace nvblox
