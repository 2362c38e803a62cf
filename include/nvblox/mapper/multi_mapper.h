// nvblox/mapper/multi_mapper.h -- nvblox::MultiMapper as constructed and driven by NvbloxNode / FuserNode
// (nvblox_node.cpp:187-210,781,1058-1062,1261-1264; fuser_node.cpp:85-94).  libnvblox_hip implements the static-TSDF
// mapping type (BASELINE.json north_star), static occupancy (nvblox_base.yaml:9) and the two human mapping types (mask-split
// depth / colour, occupancy foreground mapper) and the dynamic mapping type (freespace layer, dynamic-pixel detection, mask clean-up);
// every overload the node calls is implemented (the reference aborts on programmer errors).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <optional>
#include <vector>
#include "nvblox/mapper/mapper.h"
#include "nvblox/mapper/block_index_exchange.h"

namespace nvblox {

// MultiMapper::ground_plane_estimator() as the node's debug visualisation reads it (nvblox_node.cpp:1456,1474).  [U] restated
// (csrc/ground.hip): with multi_mapper.experimental_use_ground_plane_estimation every updateEsdf() extracts the TSDF's upward zero
// crossings whose height lies in [ground_points_candidates_min_z_m, ..._max_z_m] (nvbx_tsdf_zero_crossings) and fits a RANSAC plane
// through them (nvbx_fit_plane_ransac: ransac_distance_threshold_m, num_ransac_iterations).  Without the switch, or before the first
// update, or with fewer than three candidates, the accessors report "no estimate" and the node publishes nothing.  The plane also steers the
// 2-D ESDF slice (slice_height_above_plane_m / slice_height_thickness_m): MultiMapper::updateEsdf hands it to the mappers, Mapper::setEsdfGroundPlane.
class GroundPlaneEstimator {
 public:
  std::optional<std::vector<Vector3f>> tsdf_zero_crossings_ground_candidates() const { return candidates_; }
  std::optional<Plane> ground_plane() const { return plane_; }
  void update(nvbx_mapper* m, const MultiMapperParams& p) {
    candidates_.reset(); plane_.reset();
    const float lo = p.ground_plane_estimator_params.ground_points_candidates_min_z_m, hi = p.ground_plane_estimator_params.ground_points_candidates_max_z_m;
    int64_t n = nvbx_tsdf_zero_crossings(m, lo, hi, nullptr, 0);
    if (n < 0) checkNvbx((int)n, "nvbx_tsdf_zero_crossings");
    std::vector<float> xyz((size_t)std::max<int64_t>(n, 1) * 3);
    n = std::min<int64_t>(n, nvbx_tsdf_zero_crossings(m, lo, hi, xyz.data(), n));
    if (n <= 0) return;
    std::vector<Vector3f> pts((size_t)n);
    for (int64_t i = 0; i < n; i++) pts[(size_t)i] = Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    candidates_ = std::move(pts);
    float pl[4];
    const int64_t inl = nvbx_fit_plane_ransac(xyz.data(), n, p.ransac_plane_fitter_params.ransac_distance_threshold_m, p.ransac_plane_fitter_params.num_ransac_iterations, 1u, pl);
    if (inl >= 3) plane_ = Plane(Vector3f(pl[0], pl[1], pl[2]), pl[3]);
  }
 private:
  std::optional<std::vector<Vector3f>> candidates_;
  std::optional<Plane> plane_;
};

class MultiMapper {
 public:
  const GroundPlaneEstimator& ground_plane_estimator() const { return ground_plane_estimator_; }
  MultiMapper(float voxel_size_m, MappingType mapping_type, EsdfMode esdf_mode, MemoryType memory_type = MemoryType::kDevice,
              std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>(), int64_t block_capacity = Mapper::kDefaultBlockCapacity)
      : mapping_type_(mapping_type), esdf_mode_(esdf_mode), cuda_stream_(cuda_stream) {
    const bool occupancy_background = mapping_type == MappingType::kStaticOccupancy || mapping_type == MappingType::kHumanWithStaticOccupancy;
    human_ = mapping_type == MappingType::kHumanWithStaticTsdf || mapping_type == MappingType::kHumanWithStaticOccupancy;
    dynamic_ = mapping_type == MappingType::kDynamic;
    background_mapper_ = std::make_shared<Mapper>(voxel_size_m, memory_type,
                                                  occupancy_background ? ProjectiveLayerType::kOccupancy : (dynamic_ ? ProjectiveLayerType::kTsdfWithFreespace : ProjectiveLayerType::kTsdf),
                                                  cuda_stream, block_capacity, esdf_mode);
    // the foreground (human) mapper is an occupancy mapper (specializations/nvblox_segmentation.yaml:9-22); it is fed by the
    // masked overloads only.  In the static modes it stays empty but valid.
    foreground_mapper_ = std::make_shared<Mapper>(voxel_size_m, memory_type, ProjectiveLayerType::kOccupancy, cuda_stream, (human_ || dynamic_) ? (block_capacity ? block_capacity / 4 : (int64_t)1 << 14) : 64, esdf_mode);      // (grows on demand)
  }
  void setMapperParams(const MapperParams& background, const MapperParams& foreground) { background_mapper_->setMapperParams(background); foreground_mapper_->setMapperParams(foreground); }
  void setMapperParams(const MapperParams& params) { background_mapper_->setMapperParams(params); }   // fuser_node.cpp:94
  void setMultiMapperParams(const MultiMapperParams& p) { multi_params_ = p; }
  // nvblox_node.cpp:120: the subtree the node hangs under its own parameters and prints at start-up
  parameters::ParameterTreeNode getParameterTree() const {
    using N = parameters::ParameterTreeNode;
    return N("multi_mapper", std::vector<N>{
        N("connected_mask_component_size_threshold", multi_params_.connected_mask_component_size_threshold),
        N("remove_small_connected_components", multi_params_.remove_small_connected_components),
        background_mapper_->getParameterTree("background_mapper"), foreground_mapper_->getParameterTree("foreground_mapper")});
  }
  std::shared_ptr<Mapper> background_mapper() const { return background_mapper_; }
  std::shared_ptr<Mapper> foreground_mapper() const { return foreground_mapper_; }

  // libnvblox_hip extension (multi-GPU, one camera / one MultiMapper per GPU; block_index_exchange.h): with an exchange set, every camera
  // integrateDepth (plain, mask-split and dynamic) writes and starts this rank's block-index message, the next integrateColor (or updateEsdf, for hosts without colour) hands the
  // previous frame's gathered lists to the background mapper -- the union step before the ESDF sweep.  The node's calls do not change.
  void setBlockIndexExchange(std::shared_ptr<BlockIndexExchange> exchange) {
    if (block_index_exchange_) block_index_exchange_->drain(background_mapper_->c_handle());
    block_index_exchange_ = std::move(exchange);
  }
  const std::shared_ptr<BlockIndexExchange>& block_index_exchange() const { return block_index_exchange_; }

  void integrateDepth(const DepthImage& depth, const Transform& T_L_C, const Camera& camera, std::optional<Time> update_time_ms = std::nullopt) {
    if (!dynamic_) {
      if (block_index_exchange_) block_index_exchange_->beforeDepth(background_mapper_->c_handle());
      background_mapper_->integrateDepth(depth, T_L_C, camera);
      if (block_index_exchange_) block_index_exchange_->start(background_mapper_->c_handle());
      return;
    }
    // MappingType::kDynamic (nvblox_dynamics.yaml): pixels whose points lie in high-confidence freespace are dynamic; the mask is
    // cleaned of small components, splits the depth image; static part -> TSDF + freespace update, dynamic part -> occupancy mapper
    nvbx_mapper* m = background_mapper_->c_handle();
    const int rows = depth.rows(), cols = depth.cols();
    dynamic_mask_.resize(rows, cols); depth_background_.resize(rows, cols); depth_foreground_.resize(rows, cols); depth_overlay_.resize(rows, cols);
    float T[16]; T_L_C.toRowMajor(T);
    const float max_d = background_mapper_->tsdf_integrator().max_integration_distance_m();
    // (detect -> remove small components -> split, one call: three launches, include/nvblox_hip.h nvbx_dynamic_depth_split)
    checkNvbx(nvbx_dynamic_depth_split(m, depth.dataConstPtr(), rows, cols, T, &camera.c_abi(), max_d,
                                       multi_params_.remove_small_connected_components ? multi_params_.connected_mask_component_size_threshold : 0,
                                       multi_params_.mask_occlusion_threshold_m, dynamic_mask_.dataPtr(), depth_background_.dataPtr(), depth_foreground_.dataPtr(),
                                       reinterpret_cast<uint8_t*>(depth_overlay_.dataPtr())), "nvbx_dynamic_depth_split");
    if (update_time_ms) background_mapper_->setUpdateTime(*update_time_ms);
    if (block_index_exchange_) block_index_exchange_->beforeDepth(m);        // (the static part's blocks are what the peers' ESDF sweeps need)
    integrateDepthPair(depth_background_, depth_foreground_, T_L_C, camera);
    if (block_index_exchange_) block_index_exchange_->start(m);
    last_dynamic_T_L_C_ = T_L_C; last_dynamic_camera_ = camera;
  }
  // nvblox_node.cpp:1098,1108 (dynamic mapping): the dynamic part of the last depth frame as points of the layer frame / as an overlay
  const Pointcloud& getLastDynamicPointcloud() {
    DepthImageBackProjector bp; Pointcloud pc_C(MemoryType::kDevice);
    bp.backProjectOnGPU(depth_foreground_, last_dynamic_camera_, &pc_C, 0.0f);
    transformPointcloudOnGPU(last_dynamic_T_L_C_, pc_C, &dynamic_pointcloud_);
    return dynamic_pointcloud_;
  }
  const ColorImage& getLastDynamicFrameMaskOverlay() const { return depth_overlay_; }
  // nvblox_node.cpp:1057-1060: the depth image is split by the mask (ImageMasker::splitImageOnGPU; mask camera related to the
  // depth camera by T_CM_CD); unmasked depth -> background mapper, masked depth -> foreground (human) occupancy mapper
  void integrateDepth(const DepthImage& depth, const MonoImage& mask, const Transform& T_L_CD, const Transform& T_CM_CD, const Camera& depth_camera,
                      const Camera& mask_camera, std::optional<Time> update_time_ms = std::nullopt) {
    (void)update_time_ms;
    if (!human_) unsupported("masked depth integration outside the human mapping types");
    depth_background_.resize(depth.rows(), depth.cols()); depth_foreground_.resize(depth.rows(), depth.cols());
    depth_overlay_.resize(depth.rows(), depth.cols());
    float T[16]; T_CM_CD.toRowMajor(T);
    checkNvbx(nvbx_split_depth_by_mask(background_mapper_->c_handle(), depth.dataConstPtr(), depth.rows(), depth.cols(), mask.dataConstPtr(), mask.rows(), mask.cols(),
                                       T, &depth_camera.c_abi(), &mask_camera.c_abi(), multi_params_.mask_occlusion_threshold_m, depth_background_.dataPtr(),
                                       depth_foreground_.dataPtr(), reinterpret_cast<uint8_t*>(depth_overlay_.dataPtr())), "nvbx_split_depth_by_mask");
    if (block_index_exchange_) block_index_exchange_->beforeDepth(background_mapper_->c_handle());
    integrateDepthPair(depth_background_, depth_foreground_, T_L_CD, depth_camera);
    if (block_index_exchange_) block_index_exchange_->start(background_mapper_->c_handle());
  }
  const DepthImage& getLastDepthFrameForeground() const { return depth_foreground_; }      // nvblox_node.cpp:1126
  const DepthImage& getLastDepthFrameBackground() const { return depth_background_; }
  const ColorImage& getLastDepthFrameMaskOverlay() const { return depth_overlay_; }        // nvblox_node.cpp:1147
  // nvblox_node.cpp:1339-1384.  With use_lidar_motion_compensation the cloud carries per-point times within the scan and the node
  // supplies the sensor pose at scan end and the scan duration: the points are de-skewed into the sensor frame at scan start first.
  void integrateDepth(const Pointcloud& pointcloud, const Transform& T_L_C, const Lidar& lidar, bool use_lidar_motion_compensation = false,
                      std::optional<Transform> T_L_S_scan_end = std::nullopt, std::optional<Time> scan_duration_ms = std::nullopt,
                      std::optional<Time> update_time_ms = std::nullopt) {
    if (update_time_ms) background_mapper_->setUpdateTime(*update_time_ms);
    if (!use_lidar_motion_compensation) { background_mapper_->integrateLidarPointcloud(pointcloud, T_L_C, lidar); return; }
    if (!T_L_S_scan_end || !scan_duration_ms || !pointcloud.hasTimestamps())
      unsupported("LiDAR motion compensation without per-point timestamps / scan-end pose / scan duration");
    deskewed_.resize((size_t)pointcloud.size());
    float T0[16], T1[16]; T_L_C.toRowMajor(T0); T_L_S_scan_end->toRowMajor(T1);
    checkNvbx(nvbx_motion_compensate_pointcloud(background_mapper_->c_handle(), reinterpret_cast<const float*>(pointcloud.dataConstPtr()), pointcloud.timestampsConstPtr(),
                                                pointcloud.size(), T0, T1, (float)(int64_t)*scan_duration_ms, reinterpret_cast<float*>(deskewed_.dataPtr())),
              "nvbx_motion_compensate_pointcloud");
    background_mapper_->integrateLidarPointcloud(deskewed_, T_L_C, lidar);
  }
  const DepthImage& getLastDepthFrameFromPointcloud() const { return background_mapper_->getLastDepthFrameFromPointcloud(); }
  void integrateColor(const ColorImage& color, const Transform& T_L_C, const Camera& camera) {
    if (block_index_exchange_ && !block_index_exchange_->finishedThisFrame()) block_index_exchange_->finishPrevious(background_mapper_->c_handle(), true);
    background_mapper_->integrateColor(color, T_L_C, camera);
  }
  // nvblox_node.cpp:1261-1262: the masked pixels are removed from the colour image before the background mapper integrates it
  void integrateColor(const ColorImage& color, const MonoImage& mask, const Transform& T_L_C, const Camera& camera) {
    if (!human_) unsupported("masked colour integration outside the human mapping types");
    if (block_index_exchange_ && !block_index_exchange_->finishedThisFrame()) block_index_exchange_->finishPrevious(background_mapper_->c_handle(), true);
    color_background_.resize(color.rows(), color.cols());
    checkNvbx(nvbx_split_color_by_mask(background_mapper_->c_handle(), reinterpret_cast<const uint8_t*>(color.dataConstPtr()), color.rows(), color.cols(),
                                       mask.dataConstPtr(), reinterpret_cast<uint8_t*>(color_background_.dataPtr()), nullptr), "nvbx_split_color_by_mask");
    background_mapper_->integrateColor(color_background_, T_L_C, camera);
  }
  void updateEsdf() {
    // [U] ground-plane mode (multi_mapper.experimental_use_ground_plane_estimation, mapper_initialization.cpp:133-153): the plane is estimated from
    // the TSDF first and the 2-D slice of this update follows it (slice_height_above_plane_m / slice_height_thickness_m); no estimate = fixed heights
    if (multi_params_.experimental_use_ground_plane_estimation) {
      ground_plane_estimator_.update(background_mapper_->c_handle(), multi_params_);
      background_mapper_->setEsdfGroundPlane(ground_plane_estimator_.ground_plane());
      if (human_ || dynamic_) foreground_mapper_->setEsdfGroundPlane(ground_plane_estimator_.ground_plane());
    }
    if (block_index_exchange_ && !block_index_exchange_->finishedThisFrame()) block_index_exchange_->finishPrevious(background_mapper_->c_handle(), false);      // (a host without colour)
    background_mapper_->updateEsdf(); if (human_ || dynamic_) foreground_mapper_->updateEsdf();
  }
  void updateColorMesh(UpdateFullLayer f = UpdateFullLayer::kNo) { background_mapper_->updateColorMesh(f); }

  MappingType mapping_type() const { return mapping_type_; }
  EsdfMode esdf_mode() const { return esdf_mode_; }

 private:
  // background_mapper_->integrateDepth(bg) followed by foreground_mapper_->integrateDepth(fg): the same maps in two launches instead of four
  // (nvbx_integrate_depth_pair: the two mappers share a stream, their view-marking launches share a grid and so do their TSDF-update launches)
  void integrateDepthPair(const DepthImage& bg, const DepthImage& fg, const Transform& T_L_C, const Camera& camera) {
    timing::Timer t("tsdf/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_depth_pair(background_mapper_->c_handle(), bg.dataConstPtr(), foreground_mapper_->c_handle(), fg.dataConstPtr(), bg.rows(), bg.cols(), T, &camera.c_abi()),
              "nvbx_integrate_depth_pair");
  }
  [[noreturn]] static void unsupported(const char* what) {
    std::fprintf(stderr, "[nvblox_hip] %s is outside the MI355X hot path of this library (static TSDF + colour + 2-D ESDF + mesh)\n", what);
    std::abort();
  }
  GroundPlaneEstimator ground_plane_estimator_;
  bool human_ = false, dynamic_ = false;
  MonoImage dynamic_mask_{MemoryType::kDevice};
  Pointcloud dynamic_pointcloud_{MemoryType::kDevice}, deskewed_{MemoryType::kDevice};
  Transform last_dynamic_T_L_C_; Camera last_dynamic_camera_;
  DepthImage depth_background_{MemoryType::kDevice}, depth_foreground_{MemoryType::kDevice};
  ColorImage depth_overlay_{MemoryType::kDevice}, color_background_{MemoryType::kDevice};
  MappingType mapping_type_; EsdfMode esdf_mode_;
  std::shared_ptr<CudaStream> cuda_stream_;
  std::shared_ptr<Mapper> background_mapper_, foreground_mapper_;
  std::shared_ptr<BlockIndexExchange> block_index_exchange_;
  MultiMapperParams multi_params_;
};

}  // namespace nvblox
