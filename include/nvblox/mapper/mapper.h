// nvblox/mapper/mapper.h -- nvblox::Mapper as nvblox_ros uses it (signatures collected in SURVEY.md 8b from
// nvblox_node.cpp / layer_publishing.cpp / fuser_node.cpp), implemented as inline wrappers over the C-ABI of
// libnvblox_hip.so (include/nvblox_hip.h).  Single-caller, stream-ordered like the reference (nvblox_node.cpp:99,456-459).
#pragma once
#include <chrono>
#include <cmath>
#include <deque>
#include <memory>
#include <unordered_map>
#include <optional>
#include <string>
#include <vector>
#include "nvblox/core/cuda_stream.h"
#include "nvblox/core/types.h"
#include "nvblox/map/layer.h"
#include "nvblox/mapper/mapper_params.h"
#include "nvblox/mesh/mesh.h"
#include "nvblox/sensors/camera.h"
#include "nvblox/sensors/image.h"
#include "nvblox/sensors/lidar.h"
#include "nvblox/sensors/pointcloud.h"
#include "nvblox/utils/timing.h"
#include "nvblox_hip.h"

namespace nvblox {

// LayerType bit mask used by serializeSelectedLayers (layer_publishing.cpp:675-711)
using LayerTypeBitMask = uint32_t;
namespace LayerType {
constexpr LayerTypeBitMask kTsdf = NVBX_LAYER_TSDF, kColor = NVBX_LAYER_COLOR, kEsdf = NVBX_LAYER_ESDF, kColorMesh = NVBX_LAYER_MESH,
                           kFreespace = NVBX_LAYER_FREESPACE, kOccupancy = NVBX_LAYER_OCCUPANCY;
}
struct BlockExclusionParams {   // layer_publishing.cpp:702-707
  Vector3f exclusion_center_m; float exclusion_height_m = -1.f; float exclusion_radius_m = -1.f; float block_size_m = 0.f;
};

// SerializedLayer<VoxelType> as layer_publishing.cpp:271-339,368-377 reads it: block_indices, block_offsets (n + 1,
// in voxels), voxels (flat, 512 per block in the reference's z + 8y + 64x order)
template <typename VoxelType>
struct SerializedLayer {
  std::vector<Index3D> block_indices;
  std::vector<int> block_offsets;
  std::vector<VoxelType> voxels;
};
using SerializedTsdfLayer = SerializedLayer<TsdfVoxel>;
using SerializedColorLayer = SerializedLayer<ColorVoxel>;

// BoundingShape (sphere or AABB) as built by the EsdfAndGradients service's clearing request
// (conversions/esdf_and_gradients_conversions.cu:127-180) and passed to Mapper::clearTsdfInsideShapes (nvblox_node.cpp:1834)
struct BoundingSphere { Vector3f center; float radius = 0.f; BoundingSphere() = default; BoundingSphere(const Vector3f& c, float r) : center(c), radius(r) {} };
class BoundingShape {
 public:
  BoundingShape(const BoundingSphere& s) { c_.kind = 0; for (int i = 0; i < 3; i++) { c_.a[i] = s.center[i]; c_.b[i] = 0.f; } c_.b[0] = s.radius; }   // NOLINT
  BoundingShape(const AxisAlignedBoundingBox& b) { c_.kind = 1; for (int i = 0; i < 3; i++) { c_.a[i] = b.min()[i]; c_.b[i] = b.max()[i]; } }          // NOLINT
  const nvbx_bounding_shape& c_abi() const { return c_; }
 private:
  nvbx_bounding_shape c_{};
};

class Mapper {
 public:
  static constexpr int64_t kDefaultBlockCapacity = 0;   // automatic: ~4 % of the free HBM (2^16 .. 2^20 blocks), doubling on demand (nvbx_mapper_create)

  Mapper(float voxel_size_m, MemoryType memory_type = MemoryType::kDevice, ProjectiveLayerType projective_layer_type = ProjectiveLayerType::kTsdf,
         std::shared_ptr<CudaStream> cuda_stream = std::make_shared<CudaStreamOwning>(), int64_t block_capacity = kDefaultBlockCapacity,
         EsdfMode esdf_mode = EsdfMode::k2D)
      : voxel_size_m_(voxel_size_m), projective_layer_type_(projective_layer_type), esdf_mode_(esdf_mode), cuda_stream_(std::move(cuda_stream)) {
    (void)memory_type;   // blocks always live in HBM (MemoryType::kDevice, nvblox_node.cpp:190)
    int dev = 0; (void)hipGetDevice(&dev);
    const nvbx_mapper_params p = params_.toCAbi(voxel_size_m, projective_layer_type_, esdf_mode_);
    checkNvbx(nvbx_mapper_create(dev, (void*)(hipStream_t)(*cuda_stream_), &p, block_capacity, &m_), "nvbx_mapper_create");
    detail::contextMapper() = m_;
    rebuildViews();
  }
  ~Mapper() { if (detail::contextMapper() == m_) detail::contextMapper() = nullptr; if (m_) nvbx_mapper_destroy(m_); }
  Mapper(const Mapper&) = delete;
  Mapper& operator=(const Mapper&) = delete;

  // [U] the ground plane the 2-D ESDF slice follows (slice_height_above_plane_m / slice_height_thickness_m, mapper_initialization.cpp:257-260):
  // set by MultiMapper from its estimator before updateEsdf when multi_mapper.experimental_use_ground_plane_estimation is on; nullopt = the fixed heights
  void setEsdfGroundPlane(const std::optional<Plane>& plane) {
    nvbx_mapper_params p = params_.toCAbi(voxel_size_m_, projective_layer_type_, esdf_mode_);
    ground_plane_ = plane;
    applyGroundPlane(&p);
    checkNvbx(nvbx_mapper_set_params(m_, &p), "nvbx_mapper_set_params");
  }
  void setMapperParams(const MapperParams& params) {
    params_ = params;
    nvbx_mapper_params p = params_.toCAbi(voxel_size_m_, projective_layer_type_, esdf_mode_);
    applyGroundPlane(&p);
    checkNvbx(nvbx_mapper_set_params(m_, &p), "nvbx_mapper_set_params");
  }
  const MapperParams& params() const { return params_; }
  // nvblox_node.cpp:120 (through MultiMapper::getParameterTree)
  parameters::ParameterTreeNode getParameterTree(const std::string& name_remap = "mapper") const {
    parameters::ParameterTreeNode t = params_.getParameterTree(name_remap);
    t.children().value().insert(t.children().value().begin(), parameters::ParameterTreeNode("voxel_size_m", voxel_size_m_));
    return t;
  }

  // -- integration (asynchronous on the mapper's stream); README timer tags tsdf/integrate, color/integrate, ...
  void setUpdateTime(Time update_time_ms) { checkNvbx(nvbx_set_time_ms(m_, (int64_t)update_time_ms), "nvbx_set_time_ms"); }     // freespace layer clock
  void integrateDepth(const DepthImage& depth_frame, const Transform& T_L_C, const Camera& camera) {
    timing::Timer t("tsdf/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_depth(m_, depth_frame.dataConstPtr(), depth_frame.rows(), depth_frame.cols(), T, &camera.c_abi()), "nvbx_integrate_depth");
  }
  // The frames of up to 8 cameras of one image size as ONE launch set (nvbx_integrate_depth_batch / _color_batch): defined as equal to
  // the separate calls in order; what the node's per-tick loop over its camera queues (nvblox_node.cpp:247-292) can hand over at once.
  void integrateDepthBatch(const std::vector<const DepthImage*>& depth_frames, const std::vector<Transform>& T_L_C, const std::vector<Camera>& cameras) {
    timing::Timer t("tsdf/integrate");
    const int n = (int)depth_frames.size();
    if (n == 0) return;
    std::vector<const float*> ptr((size_t)n); std::vector<float> T((size_t)n * 16); std::vector<nvbx_camera> cams((size_t)n);
    for (int i = 0; i < n; i++) { ptr[(size_t)i] = depth_frames[(size_t)i]->dataConstPtr(); T_L_C[(size_t)i].toRowMajor(&T[(size_t)i * 16]); cams[(size_t)i] = cameras[(size_t)i].c_abi(); }
    checkNvbx(nvbx_integrate_depth_batch(m_, n, ptr.data(), depth_frames[0]->rows(), depth_frames[0]->cols(), T.data(), cams.data()), "nvbx_integrate_depth_batch");
  }
  void integrateColorBatch(const std::vector<const ColorImage*>& color_frames, const std::vector<Transform>& T_L_C, const std::vector<Camera>& cameras) {
    timing::Timer t("color/integrate");
    const int n = (int)color_frames.size();
    if (n == 0) return;
    std::vector<const uint8_t*> ptr((size_t)n); std::vector<float> T((size_t)n * 16); std::vector<nvbx_camera> cams((size_t)n);
    for (int i = 0; i < n; i++) { ptr[(size_t)i] = reinterpret_cast<const uint8_t*>(color_frames[(size_t)i]->dataConstPtr()); T_L_C[(size_t)i].toRowMajor(&T[(size_t)i * 16]); cams[(size_t)i] = cameras[(size_t)i].c_abi(); }
    checkNvbx(nvbx_integrate_color_batch(m_, n, ptr.data(), color_frames[0]->rows(), color_frames[0]->cols(), T.data(), cams.data()), "nvbx_integrate_color_batch");
  }
  // 16-bit millimetre depth: fuses conversions::depthImageFromNitrosViewAsync (image_conversions_thrust.cu:39-45)
  void integrateDepth(const Image<uint16_t>& depth_mm, const Transform& T_L_C, const Camera& camera) {
    timing::Timer t("tsdf/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_depth_u16mm(m_, depth_mm.dataConstPtr(), depth_mm.rows(), depth_mm.cols(), T, &camera.c_abi()), "nvbx_integrate_depth_u16mm");
  }
  // range image of a spinning LiDAR (rows = elevation divisions, cols = azimuth divisions, metres along the beam)
  void integrateLidarDepth(const DepthImage& range_frame, const Transform& T_L_C, const Lidar& lidar) {
    timing::Timer t("tsdf/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_lidar_depth(m_, range_frame.dataConstPtr(), range_frame.rows(), range_frame.cols(), T, &lidar.c_abi()), "nvbx_integrate_lidar_depth");
  }
  // point cloud (sensor frame) -> range image -> integration; the image stays available like the reference's
  // getLastDepthFrameFromPointcloud() (nvblox_node.cpp:1397)
  void integrateLidarPointcloud(const Pointcloud& pointcloud, const Transform& T_L_C, const Lidar& lidar) {
    last_depth_frame_from_pointcloud_.resize(lidar.rows(), lidar.cols());
    checkNvbx(nvbx_depth_image_from_pointcloud(m_, reinterpret_cast<const float*>(pointcloud.dataConstPtr()), pointcloud.size(), &lidar.c_abi(),
                                               last_depth_frame_from_pointcloud_.dataPtr()), "nvbx_depth_image_from_pointcloud");
    integrateLidarDepth(last_depth_frame_from_pointcloud_, T_L_C, lidar);
  }
  const DepthImage& getLastDepthFrameFromPointcloud() const { return last_depth_frame_from_pointcloud_; }
  void integrateColor(const ColorImage& color_frame, const Transform& T_L_C, const Camera& camera) {
    timing::Timer t("color/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_color(m_, reinterpret_cast<const uint8_t*>(color_frame.dataConstPtr()), color_frame.rows(), color_frame.cols(), T, &camera.c_abi()),
              "nvbx_integrate_color");
  }
  // bgra8 colour (4 bytes per pixel): the ToRgba<Bgra> reorder of image_conversions_thrust.cu:60-65 fused into the fetch
  void integrateColorBgra8(const Image<uint32_t>& bgra_frame, const Transform& T_L_C, const Camera& camera) {
    timing::Timer t("color/integrate");
    float T[16]; T_L_C.toRowMajor(T);
    checkNvbx(nvbx_integrate_color_bgra8(m_, reinterpret_cast<const uint8_t*>(bgra_frame.dataConstPtr()), bgra_frame.rows(), bgra_frame.cols(), T, &camera.c_abi()),
              "nvbx_integrate_color_bgra8");
  }
  void updateEsdf() { timing::Timer t("esdf/integrate"); checkNvbx(nvbx_update_esdf(m_), "nvbx_update_esdf"); }
  void updateColorMesh(UpdateFullLayer update_full_layer = UpdateFullLayer::kNo) {
    timing::Timer t("mesh/integrate");
    checkNvbx(nvbx_update_color_mesh(m_, update_full_layer == UpdateFullLayer::kYes ? 1 : 0), "nvbx_update_color_mesh");
    mesh_updates_++;
  }
  void updateMesh(UpdateFullLayer f = UpdateFullLayer::kNo) { updateColorMesh(f); }
  void decayOccupancyAllVoxels() { checkNvbx(nvbx_decay_occupancy(m_), "nvbx_decay_occupancy"); }   // nvblox_node.cpp:928 (occupancy mappers)
  void decayTsdf() { checkNvbx(nvbx_decay_tsdf(m_, 0), "nvbx_decay_tsdf"); }
  template <typename SensorType>
  void decayTsdfExcludeLastView() { checkNvbx(nvbx_decay_tsdf(m_, 1), "nvbx_decay_tsdf"); }   // nvblox_node.cpp:935
  void clearOutsideRadius(const Vector3f& center, float radius) {   // nvblox_node.cpp:1575
    checkNvbx(nvbx_clear_outside_radius(m_, center.data(), radius), "nvbx_clear_outside_radius");
  }
  void clearTsdfInsideShapes(const std::vector<BoundingShape>& shapes) {   // nvblox_node.cpp:1834
    std::vector<nvbx_bounding_shape> c; c.reserve(shapes.size());
    for (const auto& s : shapes) c.push_back(s.c_abi());
    checkNvbx(nvbx_clear_tsdf_inside_shapes(m_, c.data(), (int32_t)c.size()), "nvbx_clear_tsdf_inside_shapes");
  }
  // layer_publishing.cpp:716: blocks removed since the last call
  // (decay and radius clearing record the blocks they deallocate on the device: nvbx_take_cleared_blocks)
  std::vector<Index3D> getClearedBlocks(const std::vector<uint32_t>& = {}) {
    std::vector<Index3D> out(256);
    for (;;) {
      const int64_t n = nvbx_take_cleared_blocks(m_, reinterpret_cast<nvbx_index3d*>(out.data()), (int64_t)out.size());
      checkNvbx(n < 0 ? (int)n : 0, "nvbx_take_cleared_blocks");
      if (n <= (int64_t)out.size()) { out.resize((size_t)n); return out; }
      out.resize((size_t)n);            // list left untouched by the short call: take it with room for all of it
    }
  }
  void clear() { checkNvbx(nvbx_mapper_clear(m_), "nvbx_mapper_clear"); }

  // -- layers
  const TsdfLayer& tsdf_layer() const { return tsdf_layer_; }
  const OccupancyLayer& occupancy_layer() const { return occupancy_layer_; }       // empty unless ProjectiveLayerType::kOccupancy
  const FreespaceLayer& freespace_layer() const { return freespace_layer_; }       // empty unless ProjectiveLayerType::kTsdfWithFreespace
  const ColorLayer& color_layer() const { return color_layer_; }
  const EsdfLayer& esdf_layer() const { return esdf_layer_; }
  TsdfLayer& tsdf_layer() { return tsdf_layer_; }
  ColorLayer& color_layer() { return color_layer_; }
  EsdfLayer& esdf_layer() { return esdf_layer_; }
  float voxel_size_m() const { return voxel_size_m_; }
  ProjectiveLayerType projective_layer_type() const { return projective_layer_type_; }
  EsdfMode esdf_mode() const { return esdf_mode_; }

  // -- integrator parameter accessors used by the node (nvblox_node.cpp:136,1509,1513,1529-1533)
  struct EsdfIntegratorView {
    const Mapper* m;
    float esdf_slice_height() const { return m->params_.esdf_integrator_params.esdf_slice_height; }
    float esdf_slice_min_height() const { return m->params_.esdf_integrator_params.esdf_slice_min_height; }
    float esdf_slice_max_height() const { return m->params_.esdf_integrator_params.esdf_slice_max_height; }
    float max_esdf_distance_m() const { return m->params_.esdf_integrator_params.esdf_integrator_max_distance_m; }
  };
  struct ViewCalculatorView {
    const Mapper* m;
    WorkspaceBoundsType workspace_bounds_type() const { return m->params_.view_calculator_params.workspace_bounds_type; }
    Vector3f workspace_bounds_min_corner_m() const { const auto& v = m->params_.view_calculator_params; return {v.workspace_bounds_min_corner_x_m, v.workspace_bounds_min_corner_y_m, v.workspace_bounds_min_height_m}; }
    Vector3f workspace_bounds_max_corner_m() const { const auto& v = m->params_.view_calculator_params; return {v.workspace_bounds_max_corner_x_m, v.workspace_bounds_max_corner_y_m, v.workspace_bounds_max_height_m}; }
    int raycast_subsampling_factor() const { return m->params_.view_calculator_params.raycast_subsampling_factor; }
  };
  struct TsdfIntegratorView {
    const Mapper* m;
    ViewCalculatorView view_calculator() const { return ViewCalculatorView{m}; }
    float max_integration_distance_m() const { return m->params_.projective_integrator_params.projective_integrator_max_integration_distance_m; }
    float truncation_distance_vox() const { return m->params_.projective_integrator_params.projective_integrator_truncation_distance_vox; }
  };
  EsdfIntegratorView esdf_integrator() const { return EsdfIntegratorView{this}; }
  TsdfIntegratorView tsdf_integrator() const { return TsdfIntegratorView{this}; }
  TsdfIntegratorView occupancy_integrator() const { return TsdfIntegratorView{this}; }     // max_integration_distance_m(): nvblox_node.cpp:1130

  // -- serialization of the mesh for publishing (layer_publishing.cpp:702-711,770-776; mesh_conversions.cpp:62-104)
  // Voxel layers: every allocated TSDF block inside the exclusion cylinder (radius / height around the centre; negative =
  // unlimited) is gathered on the GPU (k_gather_blocks) and copied out; the colour layer is serialized over the SAME block
  // list, as the publisher requires (layer_publishing.cpp:316,452-459).
  // Mesh streaming is rationed like the reference's layer streamer (layer_streamer_bandwidth_limit_mbps, nvblox_base.yaml:110):
  // the blocks of every mesh update since the last call join a queue (a newer mesh of a queued block replaces it in place),
  // and each call sends the oldest blocks that fit into bandwidth_limit_mbps x (time since the last call, at most 1 s); at
  // least one block is sent so that the queue always drains.  bandwidth_limit_mbps < 0: no limit.  Every voxel-layer stream (TSDF / colour,
  // occupancy, freespace) is cut to the exclusion cylinder and rationed nearest-first (selectBlocksToStream below).
  void serializeSelectedLayers(LayerTypeBitMask layers, float bandwidth_limit_mbps = -1.f, const BlockExclusionParams& ex = BlockExclusionParams()) {
    if (layers & LayerType::kColorMesh) serializeColorMesh(bandwidth_limit_mbps);
    // every voxel-layer stream goes through the same selection: the exclusion cylinder, then the bandwidth ration (layer_publishing.cpp:702-711
    // passes layer_streamer_bandwidth_limit_mbps and the exclusion parameters for EVERY layer type in the mask)
    if (layers & LayerType::kOccupancy)
      serialized_occupancy_ = gatherLayer<OccupancyVoxel>(NVBX_LAYER_OCCUPANCY, selectBlocksToStream(occupancy_layer_.getAllBlockIndices(), occupancy_layer_.block_size(), ex,
                                                          bandwidth_limit_mbps, 12.0 + 512.0 * sizeof(OccupancyVoxel), &stream_clock_[0]));
    if (layers & LayerType::kFreespace)
      serialized_freespace_ = gatherLayer<FreespaceVoxel>(NVBX_LAYER_FREESPACE, selectBlocksToStream(freespace_layer_.getAllBlockIndices(), freespace_layer_.block_size(), ex,
                                                          bandwidth_limit_mbps, 12.0 + 512.0 * sizeof(FreespaceVoxel), &stream_clock_[1]));
    if (layers & (LayerType::kTsdf | LayerType::kColor)) {
      const double per_block = 12.0 + ((layers & LayerType::kTsdf) ? 512.0 * sizeof(TsdfVoxel) : 0.0) + ((layers & LayerType::kColor) ? 512.0 * sizeof(ColorVoxel) : 0.0);
      const std::vector<Index3D> sel = selectBlocksToStream(tsdf_layer_.getAllBlockIndices(), tsdf_layer_.block_size(), ex, bandwidth_limit_mbps, per_block, &stream_clock_[2]);
      if (layers & LayerType::kTsdf) serialized_tsdf_ = gatherLayer<TsdfVoxel>(NVBX_LAYER_TSDF, sel);
      if (layers & LayerType::kColor) serialized_color_ = gatherLayer<ColorVoxel>(NVBX_LAYER_COLOR, sel);      // (the SAME block list, as the publisher requires)
    }
  }
  std::shared_ptr<SerializedColorMeshLayer> serializedColorMeshLayer() const { return serialized_mesh_; }
  size_t numMeshBlocksAwaitingStreaming() const { return pending_order_.size(); }     // queued by the bandwidth limit
  // layer_publishing.cpp:798-822 (dynamic mapper): all occupancy blocks, refreshed by serializeSelectedLayers(LayerType::kOccupancy, ...)
  std::shared_ptr<const SerializedLayer<OccupancyVoxel>> serializedOccupancyLayer() const { return serialized_occupancy_; }
  std::shared_ptr<const SerializedLayer<FreespaceVoxel>> serializedFreespaceLayer() const { return serialized_freespace_; }     // layer_publishing.cpp:694-698
  std::shared_ptr<const SerializedTsdfLayer> serializedTsdfLayer() const { return serialized_tsdf_; }
  std::shared_ptr<const SerializedColorLayer> serializedColorLayer() const { return serialized_color_; }

  // nvblox_node.cpp:1668,1703: recoverable I/O errors are reported as `false` (nvbx_last_error() has the reason)
  bool saveLayerCake(const std::string& path) const { return nvbx_save_map(m_, path.c_str()) == NVBX_OK; }
  bool loadMap(const std::string& path) { return nvbx_load_map(m_, path.c_str()) == NVBX_OK; }

  // libnvblox_hip extension (not in the reference): cross-frame pipelining of integrateColor -- nvbx_mapper_set_color_deferral in nvblox_hip.h.
  // ON by default in its staged form (the held-back frame is copied into mapper-owned memory, so the caller's ColorImage may be refilled right after
  // integrateColor returns -- what a node does with its one colour buffer; nothing observable through this API differs from `false` but the time);
  // `zero_copy` = no copy, the caller keeps the image unchanged until the next call into the mapper (opt-in); false = the classic launch order
  void setColorIntegrationDeferred(bool on, bool zero_copy = false) { checkNvbx(nvbx_mapper_set_color_deferral(m_, on ? (zero_copy ? 1 : 2) : 0), "nvbx_mapper_set_color_deferral"); }
  void synchronize() const { checkNvbx(nvbx_synchronize(m_), "nvbx_synchronize"); }
  void flush() const { checkNvbx(nvbx_flush(m_), "nvbx_flush"); }      // enqueue held-back work without waiting
  nvbx_mapper* c_handle() const { return m_; }
  const std::shared_ptr<CudaStream>& cuda_stream() const { return cuda_stream_; }

 private:
  void applyGroundPlane(nvbx_mapper_params* p) const {
    p->esdf_use_ground_plane = ground_plane_ ? 1 : 0;
    if (ground_plane_) { p->esdf_ground_plane[0] = ground_plane_->normal().x(); p->esdf_ground_plane[1] = ground_plane_->normal().y(); p->esdf_ground_plane[2] = ground_plane_->normal().z(); p->esdf_ground_plane[3] = ground_plane_->d(); }
  }
  std::optional<Plane> ground_plane_;
  void rebuildViews() { freespace_layer_ = FreespaceLayer(m_, voxel_size_m_); occupancy_layer_ = OccupancyLayer(m_, voxel_size_m_); tsdf_layer_ = TsdfLayer(m_, voxel_size_m_); color_layer_ = ColorLayer(m_, voxel_size_m_); esdf_layer_ = EsdfLayer(m_, voxel_size_m_); }
  // Blocks of a voxel layer that go out with this call: inside the exclusion cylinder (radius / height around the centre; negative =
  // unlimited), then -- bandwidth_limit_mbps >= 0 -- [U] nearest first: ordered by distance to the exclusion centre (the robot) and cut where
  // bandwidth_limit_mbps x (time since this stream's last serialization, at most 1 s) is used up; at least one block goes out.
  struct StreamClock { std::chrono::steady_clock::time_point last; bool valid = false; };
  static std::vector<Index3D> selectBlocksToStream(const std::vector<Index3D>& all, float bs, const BlockExclusionParams& ex, float bandwidth_limit_mbps,
                                                   double bytes_per_block, StreamClock* clock) {
    std::vector<Index3D> sel;
    auto off = [&](const Index3D& b, float* dx, float* dy, float* dz) { const Vector3f c = getCenterPositionFromBlockIndex(bs, b);
      *dx = c.x() - ex.exclusion_center_m.x(); *dy = c.y() - ex.exclusion_center_m.y(); *dz = c.z() - ex.exclusion_center_m.z(); };
    for (const Index3D& b : all) {
      float dx, dy, dz; off(b, &dx, &dy, &dz);
      if (ex.exclusion_radius_m >= 0.f && dx * dx + dy * dy > ex.exclusion_radius_m * ex.exclusion_radius_m) continue;
      if (ex.exclusion_height_m >= 0.f && std::fabs(dz) > ex.exclusion_height_m) continue;
      sel.push_back(b);
    }
    if (bandwidth_limit_mbps >= 0.f && !sel.empty()) {
      const auto now = std::chrono::steady_clock::now();
      double dt = clock->valid ? std::chrono::duration<double>(now - clock->last).count() : 1.0;
      if (dt > 1.0) dt = 1.0;
      clock->last = now; clock->valid = true;
      const size_t keep = std::max<size_t>(1, (size_t)((double)bandwidth_limit_mbps * 1e6 / 8.0 * dt / bytes_per_block));
      if (keep < sel.size()) {
        auto dist2 = [&](const Index3D& b) { float dx, dy, dz; off(b, &dx, &dy, &dz); return dx * dx + dy * dy + dz * dz; };
        std::stable_sort(sel.begin(), sel.end(), [&](const Index3D& a, const Index3D& b) { return dist2(a) < dist2(b); });
        sel.resize(keep);
        std::sort(sel.begin(), sel.end());
      }
    }
    return sel;
  }
  StreamClock stream_clock_[3];          // occupancy, freespace, TSDF / colour
  template <typename VoxelType>
  std::shared_ptr<SerializedLayer<VoxelType>> gatherLayer(uint32_t layer, const std::vector<Index3D>& sel) const {
    auto out = std::make_shared<SerializedLayer<VoxelType>>();
    out->block_indices = sel;
    out->block_offsets.resize(sel.size() + 1);
    for (size_t i = 0; i <= sel.size(); i++) out->block_offsets[i] = (int)(i * 512);
    out->voxels.assign(sel.size() * 512, VoxelType());
    if (!sel.empty())   // blocks without this layer (e.g. no colour yet) stay default-initialised
      checkNvbx(nvbx_get_blocks(m_, layer, reinterpret_cast<const nvbx_index3d*>(sel.data()), (int64_t)sel.size(), out->voxels.data(), nullptr), "nvbx_get_blocks");
    return out;
  }
  struct PendingBlockMesh { std::vector<Vector3f> v, n; std::vector<Color> c; std::vector<int32_t> t; };
  void serializeColorMesh(float bandwidth_limit_mbps) {
    if (mesh_updates_ != mesh_updates_fetched_) {          // a mesh update happened since the last call: queue its blocks
      mesh_updates_fetched_ = mesh_updates_;
      auto fresh = fetchLastMeshUpdate();
      for (size_t b = 0; b < fresh->block_indices.size(); b++) {
        const Index3D& idx = fresh->block_indices[b];
        PendingBlockMesh pm;
        const size_t v0 = (size_t)fresh->vertex_block_offsets[b], v1 = (size_t)fresh->vertex_block_offsets[b + 1];
        const size_t t0 = (size_t)fresh->triangle_index_block_offsets[b], t1 = (size_t)fresh->triangle_index_block_offsets[b + 1];
        pm.v.assign(fresh->vertices.begin() + v0, fresh->vertices.begin() + v1); pm.n.assign(fresh->vertex_normals.begin() + v0, fresh->vertex_normals.begin() + v1);
        pm.c.assign(fresh->vertex_appearances.begin() + v0, fresh->vertex_appearances.begin() + v1);
        pm.t.assign(fresh->triangle_indices.begin() + t0, fresh->triangle_indices.begin() + t1);
        if (pending_mesh_.find(idx) == pending_mesh_.end()) pending_order_.push_back(idx);
        pending_mesh_[idx] = std::move(pm);
      }
    }
    const auto now = std::chrono::steady_clock::now();
    double dt = last_mesh_stream_valid_ ? std::chrono::duration<double>(now - last_mesh_stream_).count() : 1.0;
    if (dt > 1.0) dt = 1.0;
    last_mesh_stream_ = now; last_mesh_stream_valid_ = true;
    const double budget = bandwidth_limit_mbps < 0.f ? -1.0 : (double)bandwidth_limit_mbps * 1e6 / 8.0 * dt;
    auto s = std::make_shared<SerializedColorMeshLayer>();
    s->vertex_block_offsets.push_back(0); s->triangle_index_block_offsets.push_back(0);
    double sent = 0.0;
    while (!pending_order_.empty()) {
      const Index3D idx = pending_order_.front();
      const PendingBlockMesh& pm = pending_mesh_[idx];
      const double bytes = 12.0 + (double)pm.v.size() * (12 + 12 + 3) + (double)pm.t.size() * 4;
      if (budget >= 0.0 && sent > 0.0 && sent + bytes > budget) break;
      sent += bytes;
      s->block_indices.push_back(idx);
      s->vertices.insert(s->vertices.end(), pm.v.begin(), pm.v.end()); s->vertex_normals.insert(s->vertex_normals.end(), pm.n.begin(), pm.n.end());
      s->vertex_appearances.insert(s->vertex_appearances.end(), pm.c.begin(), pm.c.end());
      s->triangle_indices.insert(s->triangle_indices.end(), pm.t.begin(), pm.t.end());
      s->vertex_block_offsets.push_back((int32_t)s->vertices.size()); s->triangle_index_block_offsets.push_back((int32_t)s->triangle_indices.size());
      pending_mesh_.erase(idx); pending_order_.pop_front();
    }
    serialized_mesh_ = s;
  }
  std::shared_ptr<SerializedColorMeshLayer> fetchLastMeshUpdate() {
    int64_t nb = 0, nv = 0, nt = 0;
    checkNvbx(nvbx_mesh_sizes(m_, &nb, &nv, &nt), "nvbx_mesh_sizes");
    auto s = std::make_shared<SerializedColorMeshLayer>();
    s->block_indices.resize((size_t)nb); s->vertices.resize((size_t)nv); s->vertex_normals.resize((size_t)nv);
    s->vertex_appearances.resize((size_t)nv); s->triangle_indices.resize((size_t)nt * 3);
    s->vertex_block_offsets.assign((size_t)nb + 1, 0); s->triangle_index_block_offsets.assign((size_t)nb + 1, 0);
    std::vector<uint8_t> rgba((size_t)nv * 4);
    checkNvbx(nvbx_mesh_copy(m_, reinterpret_cast<nvbx_index3d*>(s->block_indices.data()), s->vertex_block_offsets.data(),
                             s->triangle_index_block_offsets.data(), reinterpret_cast<float*>(s->vertices.data()),
                             reinterpret_cast<float*>(s->vertex_normals.data()), rgba.data(), s->triangle_indices.data()), "nvbx_mesh_copy");
    for (size_t i = 0; i < (size_t)nv; i++) s->vertex_appearances[i] = Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]);
    for (auto& o : s->triangle_index_block_offsets) o *= 3;   // triangles -> indices
    return s;
  }

  float voxel_size_m_;
  ProjectiveLayerType projective_layer_type_;
  EsdfMode esdf_mode_ = EsdfMode::k2D;
  std::shared_ptr<CudaStream> cuda_stream_;
  MapperParams params_;
  nvbx_mapper* m_ = nullptr;
  TsdfLayer tsdf_layer_; FreespaceLayer freespace_layer_; OccupancyLayer occupancy_layer_; ColorLayer color_layer_; EsdfLayer esdf_layer_;
  uint64_t mesh_updates_ = 0, mesh_updates_fetched_ = 0;
  std::unordered_map<Index3D, PendingBlockMesh, Index3DHash> pending_mesh_; std::deque<Index3D> pending_order_;
  std::chrono::steady_clock::time_point last_mesh_stream_{}; bool last_mesh_stream_valid_ = false;
  DepthImage last_depth_frame_from_pointcloud_{MemoryType::kDevice};
  std::shared_ptr<SerializedColorMeshLayer> serialized_mesh_ = std::make_shared<SerializedColorMeshLayer>();
  std::shared_ptr<SerializedTsdfLayer> serialized_tsdf_ = std::make_shared<SerializedTsdfLayer>();
  std::shared_ptr<SerializedLayer<OccupancyVoxel>> serialized_occupancy_ = std::make_shared<SerializedLayer<OccupancyVoxel>>();
  std::shared_ptr<SerializedLayer<FreespaceVoxel>> serialized_freespace_ = std::make_shared<SerializedLayer<FreespaceVoxel>>();
  std::shared_ptr<SerializedColorLayer> serialized_color_ = std::make_shared<SerializedColorLayer>();
};

}  // namespace nvblox
