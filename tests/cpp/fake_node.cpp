// fake_node.cpp -- a ROS-free translation unit that drives libnvblox_hip through the nvblox:: C++ facade with the SAME call
// expressions nvblox_ros uses (reference lines quoted at each call).  It proves the drop-in boundary compiles and links
// (CPU test) and, on a GPU box, that the facade produces the same map as the ctypes path the parity tests use.
//
// usage: fake_node <frames.bin> [mesh]      prints one JSON line.
// frames.bin: int32 n, rows, cols; float fu, fv, cu, cv; then n x { float T_L_C[16] row-major, float depth[rows*cols], uint8 rgb[rows*cols*3] }
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <optional>
#include <string>
#include <vector>
#include <nvblox/nvblox.h>

using namespace nvblox;

// node_params.hpp:84 + nvblox_node.cpp:90-95: node-level parameters are nvblox::Param<T> built from constexpr descriptions
constexpr Param<float>::Description kVoxelSizeParamDesc{"voxel_size", .05F, "Voxel size (in meters) to use for the map."};
constexpr StringParam::Description kGlobalFrameParamDesc{"global_frame", "odom", "The name of the TF frame to be used as the global frame."};

// mapper_initialization.cpp:156-229 declareMapperParameters / utils.hpp:60-82: what the ROS layer does with a core description
template <typename T>
static std::string declareParameter(const std::string& name_prefix, const typename Param<T>::Description& desc) {
  return name_prefix + "." + desc.name + " = " + std::to_string(desc.default_value) + "  # " + desc.help_string;
}

struct FakeNode {
  // nvblox_node.hpp:470-488,539,548: members of NvbloxNode
  std::shared_ptr<CudaStream> cuda_stream_;
  std::shared_ptr<MultiMapper> multi_mapper_;
  std::shared_ptr<Mapper> static_mapper_, dynamic_mapper_;
  DepthImage depth_image_{MemoryType::kDevice};
  ColorImage color_image_{MemoryType::kDevice};
  EsdfSlicer esdf_slicer_;
  Transform T_L_C_depth_;
  struct { float voxel_size = 0.05f; MappingType mapping_type = MappingType::kStaticTsdf; EsdfMode esdf_mode = EsdfMode::k2D;
           float distance_map_unknown_value_optimistic = 1000.0f; } params_;

  Param<float> voxel_size{kVoxelSizeParamDesc};
  StringParam global_frame{kGlobalFrameParamDesc};

  FakeNode() {
    params_.voxel_size = voxel_size.get();
    if (global_frame.get() != "odom" ||
        declareParameter<float>("static_mapper", kEsdfSliceHeightParamDesc).find("static_mapper.esdf_slice_height") != 0 ||
        declareParameter<int>("static_mapper", kDepthPreprocessingNumDilationsParamDesc).empty()) { std::fprintf(stderr, "param descriptions broken\n"); std::exit(1); }
    // nvblox_node.cpp:91
    cuda_stream_ = CudaStream::createCudaStream(static_cast<CudaStreamType>(2));
    // nvblox_node.cpp:186-190
    multi_mapper_ =
      std::make_shared<MultiMapper>(
      params_.voxel_size, params_.mapping_type, params_.esdf_mode,
      MemoryType::kDevice, cuda_stream_);
    // what getMapperParamsFromROS("static_mapper") yields for fuser.yaml (mapper_initialization.cpp:231-466)
    MapperParams static_mapper_params, dynamic_mapper_params;
    auto& pi = static_mapper_params.projective_integrator_params;
    pi.projective_integrator_max_integration_distance_m = 8.0f;
    pi.projective_integrator_truncation_distance_vox = 4.0f;
    pi.projective_integrator_weighting_mode = WeightingFunctionType::kConstantWeight;
    pi.projective_integrator_max_weight = 5.0f;
    static_mapper_params.view_calculator_params.raycast_subsampling_factor = 4;
    static_mapper_params.esdf_integrator_params.esdf_integrator_min_weight = 0.1f;
    static_mapper_params.esdf_integrator_params.esdf_integrator_max_site_distance_vox = 2.0f;
    static_mapper_params.esdf_integrator_params.esdf_integrator_max_distance_m = 2.0f;
    static_mapper_params.esdf_integrator_params.esdf_slice_height = 0.09f;
    static_mapper_params.esdf_integrator_params.esdf_slice_min_height = 0.09f;
    static_mapper_params.esdf_integrator_params.esdf_slice_max_height = 0.65f;
    static_mapper_params.mesh_integrator_params.mesh_integrator_min_weight = 0.1f;
    MultiMapperParams multi_mapper_params;
    // nvblox_node.cpp:203-204
    multi_mapper_->setMapperParams(static_mapper_params, dynamic_mapper_params);
    multi_mapper_->setMultiMapperParams(multi_mapper_params);
    // nvblox_node.cpp:209-210
    static_mapper_ = multi_mapper_.get()->background_mapper();
    dynamic_mapper_ = multi_mapper_.get()->foreground_mapper();
  }

  // nvblox_node.cpp:974-1091 processDepthImage, minus ROS
  bool processDepthImage(const float* depth_host, int rows, int cols, const Transform& T_L_C, const Camera& depth_camera_) {
    T_L_C_depth_ = T_L_C;
    depth_image_.copyFromAsync(rows, cols, depth_host, *cuda_stream_);      // image_conversions.cpp:147-155
    const Time update_time_ms(0);
    multi_mapper_->integrateDepth(depth_image_, T_L_C_depth_, depth_camera_, update_time_ms);   // :1062
    return true;
  }
  // nvblox_node.cpp:1186-1275 processColorImage
  bool processColorImage(const Color* color_host, int rows, int cols, const Transform& T_L_C, const Camera& color_camera) {
    color_image_.copyFromAsync(rows, cols, color_host, *cuda_stream_);
    multi_mapper_->integrateColor(color_image_, T_L_C, color_camera);       // :1264
    return true;
  }
  // nvblox_node.cpp:774-817 processEsdf + :819-889 sliceAndPublishEsdf
  void processEsdf(std::vector<float>* slice_host, int* width, int* height, AxisAlignedBoundingBox* aabb_out, std::vector<int8_t>* occupancy) {
    multi_mapper_->updateEsdf();                                            // :781
    const std::shared_ptr<Mapper>& mapper = static_mapper_;
    const float unknown_value = params_.distance_map_unknown_value_optimistic;
    AxisAlignedBoundingBox aabb;
    Image<float> map_slice_image(MemoryType::kDevice);
    esdf_slicer_.sliceLayerToDistanceImage(
      mapper->esdf_layer(),
      mapper->esdf_integrator().esdf_slice_height(), unknown_value,
      &aabb, &map_slice_image);                                             // :841-844
    // esdf_slice_conversions.cu:86-108 distanceMapSliceMsgFromSliceImage
    *width = map_slice_image.cols();
    *height = map_slice_image.rows();
    slice_host->assign((size_t)(*width) * (*height), unknown_value);
    (void)hipMemcpyAsync(slice_host->data(), map_slice_image.dataConstPtr(), map_slice_image.numel() * sizeof(float), hipMemcpyDefault, *cuda_stream_);
    cuda_stream_->synchronize();
    *aabb_out = aabb;
    // nvblox_node.cpp:914-919 publishOccupancyGridMsg
    occupancy->assign((size_t)(*width) * (*height), (int8_t)-1);
    esdf_slicer_.occupancyGridFromSliceImage(map_slice_image, occupancy->data(), unknown_value);
  }
  // layer_publishing.cpp:686-711 + mesh_conversions.cpp:62-104
  void publishMesh(size_t* n_blocks, size_t* n_vertices, size_t* n_triangle_indices, double* vertex_sum) {
    std::shared_ptr<Mapper> static_mapper = static_mapper_;
    static_mapper->updateColorMesh();                                       // :688
    BlockExclusionParams block_exclusion_params{
      .exclusion_center_m = T_L_C_depth_.translation(),
      .exclusion_height_m = -1.f,
      .exclusion_radius_m = -1.f,
      .block_size_m = static_mapper->tsdf_layer().block_size(),
    };
    static_mapper->serializeSelectedLayers(LayerType::kColorMesh, -1.f, block_exclusion_params);   // :709-711
    const std::vector<Index3D> blocks_to_remove_static_mapper = static_mapper->getClearedBlocks({});   // :715-716
    (void)blocks_to_remove_static_mapper;
    const std::shared_ptr<SerializedColorMeshLayer> serialized_mesh = static_mapper->serializedColorMeshLayer();
    const size_t num_blocks = serialized_mesh->block_indices.size();
    *n_blocks = num_blocks; *n_vertices = 0; *n_triangle_indices = 0; *vertex_sum = 0.0;
    for (size_t i_block = 0; i_block < num_blocks; ++i_block) {
      const int num_vertices = serialized_mesh->getNumVerticesInBlock(i_block);
      const int num_triangle_indices = serialized_mesh->getNumTriangleIndicesInBlock(i_block);
      for (size_t i_vert = 0; i_vert < static_cast<size_t>(num_vertices); ++i_vert) {
        const Vector3f v = serialized_mesh->getVertex(i_block, i_vert);
        const Color c = serialized_mesh->getAppearance(i_block, i_vert);
        *vertex_sum += (double)v.x() + (double)v.y() + (double)v.z() + (double)c.r;
      }
      for (size_t i_tri = 0; i_tri < static_cast<size_t>(num_triangle_indices); ++i_tri) {
        if (serialized_mesh->getTriangleIndex(i_block, i_tri) >= num_vertices) { std::fprintf(stderr, "bad triangle index\n"); std::exit(1); }
      }
      *n_vertices += (size_t)num_vertices; *n_triangle_indices += (size_t)num_triangle_indices;
    }
  }
};

// nvblox_node.cpp:1277-1420 processLidarPointcloud, minus ROS: model, point cloud on the device, integrate, range image out
static int lidarMain() {
  const int lidar_width = 256, lidar_height = 16; const float lidar_min_valid_range_m = 0.1f, lidar_vertical_fov_rad = 0.5236f;
  std::shared_ptr<CudaStream> cuda_stream_ = CudaStream::createCudaStream(static_cast<CudaStreamType>(2));
  auto multi_mapper_ = std::make_shared<MultiMapper>(0.1f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, cuda_stream_);
  MapperParams p; p.projective_integrator_params.lidar_projective_integrator_max_integration_distance_m = 15.0f;   // nvblox_os1.yaml:31
  p.projective_integrator_params.projective_integrator_weighting_mode = WeightingFunctionType::kConstantWeight;
  p.view_calculator_params.raycast_subsampling_factor = 2;                                                         // nvblox_os1.yaml:33
  multi_mapper_->setMapperParams(p);
  Lidar lidar = Lidar(lidar_width, lidar_height, lidar_min_valid_range_m, lidar_vertical_fov_rad);                  // :1315-1323
  // one return per beam from the inside of a sphere of radius 6 m
  std::vector<Vector3f> pts;
  for (int k = 0; k < lidar_height; k++) for (int j = 0; j < lidar_width; j++) {
    const float el = 0.5f * lidar_vertical_fov_rad - k * (lidar_vertical_fov_rad / (lidar_height - 1)), az = -3.14159265f + j * (6.2831853f / lidar_width);
    pts.emplace_back(6.f * std::cos(el) * std::cos(az), 6.f * std::cos(el) * std::sin(az), 6.f * std::sin(el));
  }
  Pointcloud nvblox_pointcloud(MemoryType::kDevice);
  nvblox_pointcloud.copyFromAsync(pts, *cuda_stream_);
  const Transform T_L_C = Transform::Identity();
  const bool use_lidar_motion_compensation = false;
  std::optional<Transform> maybe_T_L_S_scanEnd; std::optional<Time> maybe_scan_duration_ms; const Time update_time_ms(0);
  multi_mapper_->integrateDepth(nvblox_pointcloud, T_L_C, lidar,
                                use_lidar_motion_compensation, maybe_T_L_S_scanEnd,
                                maybe_scan_duration_ms, update_time_ms);                                             // :1382-1384
  const DepthImage& range = multi_mapper_->getLastDepthFrameFromPointcloud();                                     // :1397
  std::vector<float> host((size_t)range.numel());
  (void)hipMemcpyAsync(host.data(), range.dataConstPtr(), host.size() * sizeof(float), hipMemcpyDefault, *cuda_stream_);
  cuda_stream_->synchronize();
  size_t valid = 0; double sum = 0.0;
  for (float v : host) if (v > 0.f) { valid++; sum += v; }
  // the same scan with use_lidar_motion_compensation (nvblox_node.cpp:1339-1384): per-point times, pose at scan end, scan duration;
  // with identical start / end poses the de-skewed cloud is the cloud itself
  {
    auto mm2 = std::make_shared<MultiMapper>(0.1f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, cuda_stream_);
    mm2->setMapperParams(p);
    std::vector<float> rel_ms(pts.size());
    for (size_t i = 0; i < pts.size(); i++) rel_ms[i] = 100.0f * (float)i / (float)pts.size();
    nvblox_pointcloud.copyTimestampsFromAsync(rel_ms.data(), rel_ms.size(), *cuda_stream_);
    mm2->integrateDepth(nvblox_pointcloud, T_L_C, lidar, true, std::optional<Transform>(T_L_C), std::optional<Time>(Time(100)), update_time_ms);
    cuda_stream_->synchronize();
    if (mm2->background_mapper()->tsdf_layer().numAllocatedBlocks() != multi_mapper_->background_mapper()->tsdf_layer().numAllocatedBlocks()) {
      std::fprintf(stderr, "motion-compensated scan gave a different map\n"); return 1; }
  }
  std::printf("{\"lidar_blocks\": %d, \"range_valid\": %zu, \"range_mean\": %.6f}\n",
              multi_mapper_->background_mapper()->tsdf_layer().numAllocatedBlocks(), valid, valid ? sum / valid : 0.0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "lidar") return lidarMain();
  if (argc < 2) { std::fprintf(stderr, "usage: %s frames.bin | lidar\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t hdr[3]; float k[4];
  if (std::fread(hdr, 4, 3, f) != 3 || std::fread(k, 4, 4, f) != 4) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  warmupCuda();                                                             // fuser_node_main.cpp:38
  FakeNode node;
  {   // nvblox_node.cpp:119-124: the mapper's parameter subtree hangs under the node's own and is printed at start-up
    parameters::ParameterTreeNode parameter_tree_{"nvblox_node", {}};
    parameter_tree_.children().value().push_back(node.multi_mapper_->getParameterTree());
    const std::string txt = parameters::parameterTreeToString(parameter_tree_);
    if (txt.find("  multi_mapper:\n") == std::string::npos || txt.find("background_mapper:") == std::string::npos ||
        txt.find("esdf_slice_height: 0.09") == std::string::npos) { std::fprintf(stderr, "parameter tree:\n%s", txt.c_str()); return 1; }
  }
  const Camera camera(k[0], k[1], k[2], k[3], cols, rows);                  // image_conversions.cpp:27-32
  std::vector<float> depth((size_t)rows * cols); std::vector<Color> rgb((size_t)rows * cols); float T[16];
  for (int i = 0; i < n; i++) {
    timing::Rates::tick("ros/tick");                                      // nvblox_node.cpp:588
    if (std::fread(T, 4, 16, f) != 16 || std::fread(depth.data(), 4, depth.size(), f) != depth.size() || std::fread(rgb.data(), 3, rgb.size(), f) != rgb.size()) return 2;
    const Transform T_L_C = Transform::fromRowMajor(T);
    node.processDepthImage(depth.data(), rows, cols, T_L_C, camera);
    node.processColorImage(rgb.data(), rows, cols, T_L_C, camera);
    node.cuda_stream_->synchronize();      // host staging buffers are reused next iteration
  }
  std::fclose(f);
  std::vector<float> slice; std::vector<int8_t> occ; int width = 0, height = 0; AxisAlignedBoundingBox aabb;
  node.processEsdf(&slice, &width, &height, &aabb, &occ);
  size_t mb = 0, mv = 0, mt = 0; double vsum = 0.0;
  node.publishMesh(&mb, &mv, &mt, &vsum);
  // layer_publishing.cpp:696-711,744-763: colour streaming needs the TSDF layer too, both over the same block list
  LayerTypeBitMask layers_to_serialize = LayerType::kColor;
  if (layers_to_serialize & LayerType::kColor) { layers_to_serialize |= LayerType::kTsdf; }
  BlockExclusionParams block_exclusion_params{
    .exclusion_center_m = node.T_L_C_depth_.translation(), .exclusion_height_m = -1.f, .exclusion_radius_m = 2.0f,
    .block_size_m = node.static_mapper_->tsdf_layer().block_size()};
  node.static_mapper_->serializeSelectedLayers(layers_to_serialize, -1.f, block_exclusion_params);
  const auto st = node.static_mapper_->serializedTsdfLayer(); const auto sc = node.static_mapper_->serializedColorLayer();
  if (st->block_indices.size() != sc->block_indices.size()) { std::fprintf(stderr, "layer block lists differ\n"); return 1; }
  size_t ser_visible = 0;
  for (size_t i = 0; i < st->block_indices.size(); i++) {
    const int offset_layer1 = st->block_offsets[i], offset_layer2 = sc->block_offsets[i];
    for (int lin_index = 0; lin_index < st->block_offsets[i + 1] - offset_layer1; lin_index++) {
      const TsdfVoxel& v1 = st->voxels[offset_layer1 + lin_index]; const ColorVoxel& v2 = sc->voxels[offset_layer2 + lin_index];
      if (v1.weight > 0.1f && std::fabs(v1.distance) < 0.05f && v2.weight > 0.f) ser_visible++;      // the TSDF visibility filter (:142-193)
    }
  }
  // summary the Python test compares with the ctypes path
  const TsdfLayer& tsdf = node.static_mapper_->tsdf_layer();
  double tsdf_sum = 0.0; size_t observed = 0;
  callFunctionOnAllVoxels<TsdfVoxel>(tsdf, [&](const Index3D&, const Index3D&, const TsdfVoxel* v) { if (v->weight > 0.f) { tsdf_sum += (double)v->distance * (double)v->weight; observed++; } });
  double slice_sum = 0.0; size_t known = 0, occupied = 0;
  for (size_t i = 0; i < slice.size(); i++) { if (slice[i] < 999.0f) { slice_sum += slice[i]; known++; } if (occ[i] == 100) occupied++; }
  // save_ply service (nvblox_node.cpp:1609-1613) and the EsdfAndGradients clearing request (nvblox_node.cpp:1834)
  node.static_mapper_->updateColorMesh(UpdateFullLayer::kYes);
  node.static_mapper_->serializeSelectedLayers(LayerType::kColorMesh);
  const bool ply_ok = io::outputColorMeshLayerToPly(*node.static_mapper_->serializedColorMeshLayer(), std::string(argv[1]) + ".ply");
  std::vector<BoundingShape> shapes_to_clear;
  shapes_to_clear.push_back(BoundingShape(BoundingSphere(Vector3f(100.f, 100.f, 100.f), 0.5f)));       // far away: a no-op on this map
  shapes_to_clear.push_back(BoundingShape(AxisAlignedBoundingBox(Vector3f(90.f, 90.f, 90.f), Vector3f(91.f, 91.f, 91.f))));
  node.static_mapper_->clearTsdfInsideShapes(shapes_to_clear);
  if (!ply_ok) { std::fprintf(stderr, "ply export failed\n"); return 1; }
  // back-projected depth cloud (fuser_node.cpp:291-297) and the two-mapper costmap slice (nvblox_node.cpp:836-840)
  {
    DepthImageBackProjector image_back_projector_;
    Pointcloud pointcloud_C_device_(MemoryType::kDevice), pointcloud_L_device_(MemoryType::kDevice);
    DepthImage flat(120, 160, MemoryType::kDevice);
    std::vector<float> host_depth(120 * 160, 1.5f); host_depth[0] = 0.0f;
    flat.copyFromAsync(120, 160, host_depth.data(), CudaStreamOwning());
    const Camera depth_camera(80.f, 80.f, 79.5f, 59.5f, 160, 120);
    image_back_projector_.backProjectOnGPU(flat, depth_camera, &pointcloud_C_device_, 5.0f);
    transformPointcloudOnGPU(Transform::Identity(), pointcloud_C_device_, &pointcloud_L_device_);
    node.static_mapper_->synchronize();
    if (pointcloud_C_device_.size() != 120 * 160 - 1 || pointcloud_L_device_.size() != pointcloud_C_device_.size()) {
      std::fprintf(stderr, "back projection gave %d points\n", pointcloud_C_device_.size()); return 1; }
    EsdfSlicer esdf_slicer_;
    AxisAlignedBoundingBox aabb2; Image<float> combined(MemoryType::kDevice);
    esdf_slicer_.sliceLayersToCombinedDistanceImage(node.static_mapper_->esdf_layer(), node.static_mapper_->esdf_layer(), 0.09f, 0.09f, 1000.0f, &aabb2, &combined);
    if (combined.rows() != height || combined.cols() != width) { std::fprintf(stderr, "combined slice size differs\n"); return 1; }
  }
  // ground plane estimation (nvblox_node.cpp:1456,1474; multi_mapper.experimental_use_ground_plane_estimation, mapper_initialization.cpp:133-153):
  // a camera 1 m above a floor, looking down: candidates on the floor, plane z = 0
  {
    MultiMapper gp(0.05f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, std::make_shared<CudaStreamOwning>(), 1 << 12);
    MultiMapperParams mp; mp.experimental_use_ground_plane_estimation = true;
    mp.ground_plane_estimator_params.ground_points_candidates_min_z_m = -0.2f; mp.ground_plane_estimator_params.ground_points_candidates_max_z_m = 0.2f;
    mp.ransac_plane_fitter_params.ransac_distance_threshold_m = 0.03f; mp.ransac_plane_fitter_params.num_ransac_iterations = 200;
    gp.setMultiMapperParams(mp);
    if (gp.ground_plane_estimator().ground_plane() || gp.ground_plane_estimator().tsdf_zero_crossings_ground_candidates()) { std::fprintf(stderr, "ground plane before any update\n"); return 1; }
    DepthImage down(120, 160, MemoryType::kDevice);
    std::vector<float> host_depth(120 * 160, 1.0f);                       // optical axis = -z of the world: depth 1 m everywhere = the plane z = 0
    down.copyFromAsync(120, 160, host_depth.data(), CudaStreamOwning());
    Transform T_L_C = Transform::Identity();
    T_L_C(1, 1) = -1.f; T_L_C(2, 2) = -1.f;                              // camera x = world x, y = -y, z = -z (looking down)
    T_L_C.setTranslation(Vector3f(0.f, 0.f, 1.0f));
    gp.integrateDepth(down, T_L_C, Camera(80.f, 80.f, 79.5f, 59.5f, 160, 120));
    gp.updateEsdf();
    const auto cand = gp.ground_plane_estimator().tsdf_zero_crossings_ground_candidates();
    const auto plane = gp.ground_plane_estimator().ground_plane();
    if (!cand || cand->size() < 500 || !plane || std::fabs(plane->normal().z() - 1.f) > 1e-3f || std::fabs(plane->d()) > 0.02f ||
        std::fabs(plane->getHeightAtXY(Vector2f(0.2f, -0.1f))) > 0.02f) {
      std::fprintf(stderr, "ground plane: %zu candidates, normal z %g, d %g\n", cand ? cand->size() : (size_t)0, plane ? plane->normal().z() : 0.f, plane ? plane->d() : 0.f); return 1; }
    float zmin = 1e9f, zmax = -1e9f;
    for (const Vector3f& q : *cand) { zmin = std::min(zmin, q.z()); zmax = std::max(zmax, q.z()); }
    if (zmin < -0.03f || zmax > 0.03f) { std::fprintf(stderr, "ground candidates off the floor: z %g .. %g\n", zmin, zmax); return 1; }
  }
  // mapping_type "static_occupancy" (nvblox_base.yaml:9): an occupancy MultiMapper fed the same way; decayOccupancyAllVoxels
  {
    MultiMapper occ(0.05f, MappingType::kStaticOccupancy, EsdfMode::k2D, MemoryType::kDevice, std::make_shared<CudaStreamOwning>(), 1 << 12);
    DepthImage flat(120, 160, MemoryType::kDevice);
    std::vector<float> host_depth(120 * 160, 1.5f);
    flat.copyFromAsync(120, 160, host_depth.data(), CudaStreamOwning());
    const Camera depth_camera(80.f, 80.f, 79.5f, 59.5f, 160, 120);
    occ.integrateDepth(flat, Transform::Identity(), depth_camera);
    occ.updateEsdf();
    std::shared_ptr<Mapper> dynamic_mapper_ = occ.background_mapper();
    const int occ_blocks = dynamic_mapper_->occupancy_layer().numAllocatedBlocks();
    size_t occupied_voxels = 0;
    callFunctionOnAllVoxels<OccupancyVoxel>(dynamic_mapper_->occupancy_layer(), [&](const Index3D&, const Index3D&, const OccupancyVoxel* v) { if (v->log_odds > 1e-3f) occupied_voxels++; });
    dynamic_mapper_->decayOccupancyAllVoxels();
    dynamic_mapper_->serializeSelectedLayers(LayerType::kOccupancy);
    if (occ_blocks < 10 || occupied_voxels < 100 || dynamic_mapper_->tsdf_layer().numAllocatedBlocks() != 0 ||
        dynamic_mapper_->serializedOccupancyLayer()->block_indices.size() != (size_t)dynamic_mapper_->occupancy_layer().numAllocatedBlocks() ||
        !(dynamic_mapper_->occupancy_integrator().max_integration_distance_m() > 0.f)) {
      std::fprintf(stderr, "occupancy mapper: %d blocks, %zu occupied voxels\n", occ_blocks, occupied_voxels); return 1; }
    // the decay switches a node can set (mapper_initialization.cpp:383-428) are honoured, not refused: occupied voxels decay to FREE and
    // fully decayed blocks stay allocated
    MapperParams sw;
    sw.occupancy_decay_integrator_params.occupancy_decay_to_free = true;
    sw.occupancy_decay_integrator_params.occupied_region_decay_probability = 0.05f;
    sw.decay_integrator_base_params.decay_integrator_deallocate_decayed_blocks = false;
    sw.tsdf_decay_integrator_params.tsdf_set_free_distance_on_decayed = true;
    occ.setMapperParams(sw);
    const int before = dynamic_mapper_->occupancy_layer().numAllocatedBlocks();
    for (int k = 0; k < 4; k++) dynamic_mapper_->decayOccupancyAllVoxels();
    size_t still_occupied = 0, now_free = 0;
    callFunctionOnAllVoxels<OccupancyVoxel>(dynamic_mapper_->occupancy_layer(), [&](const Index3D&, const Index3D&, const OccupancyVoxel* v) {
      if (v->log_odds > 0.f) still_occupied++;
      if (v->log_odds < 0.f) now_free++; });
    if (dynamic_mapper_->occupancy_layer().numAllocatedBlocks() != before || still_occupied != 0 || now_free < occupied_voxels) {
      std::fprintf(stderr, "decay switches: %d -> %d blocks, %zu occupied, %zu free\n", before, dynamic_mapper_->occupancy_layer().numAllocatedBlocks(), still_occupied, now_free); return 1; }
  }
  // mapping_type "human_with_static_tsdf" (specializations/nvblox_segmentation.yaml): the masked overloads of nvblox_node.cpp:1057-1060,1261-1262
  {
    auto multi_mapper_ = std::make_shared<MultiMapper>(0.05f, MappingType::kHumanWithStaticTsdf, EsdfMode::k2D, MemoryType::kDevice,
                                                       std::make_shared<CudaStreamOwning>(), 1 << 12);
    DepthImage depth_image_(120, 160, MemoryType::kDevice); MonoImage mask_image_(120, 160, MemoryType::kDevice); ColorImage color_image_(120, 160, MemoryType::kDevice);
    std::vector<float> host_depth(120 * 160, 2.0f); std::vector<uint8_t> host_mask(120 * 160, 0); std::vector<Color> host_color(120 * 160, Color(200, 100, 50));
    for (int r = 40; r < 100; r++) for (int c = 60; c < 100; c++) { host_mask[r * 160 + c] = 255; host_depth[r * 160 + c] = 1.2f; }      // a person 1.2 m away
    depth_image_.copyFromAsync(120, 160, host_depth.data(), CudaStreamOwning());
    mask_image_.copyFromAsync(120, 160, host_mask.data(), CudaStreamOwning());
    color_image_.copyFromAsync(120, 160, host_color.data(), CudaStreamOwning());
    const Camera depth_camera_(80.f, 80.f, 79.5f, 59.5f, 160, 120), mask_camera(80.f, 80.f, 79.5f, 59.5f, 160, 120);
    const Transform T_L_C_depth_ = Transform::Identity(), T_CM_CD = Transform::Identity();
    multi_mapper_->integrateDepth(depth_image_, mask_image_, T_L_C_depth_, T_CM_CD, depth_camera_, mask_camera);
    multi_mapper_->integrateColor(color_image_, mask_image_, T_L_C_depth_, depth_camera_);
    multi_mapper_->updateEsdf();
    const DepthImage& depth_image_only_humans = multi_mapper_->getLastDepthFrameForeground();
    DepthImageBackProjector image_back_projector_; Pointcloud human_pointcloud_C_device_(MemoryType::kDevice);
    image_back_projector_.backProjectOnGPU(depth_image_only_humans, depth_camera_, &human_pointcloud_C_device_,
                                           multi_mapper_->foreground_mapper()->occupancy_integrator().max_integration_distance_m());
    const int human_blocks = multi_mapper_->foreground_mapper()->occupancy_layer().numAllocatedBlocks();
    const int static_blocks = multi_mapper_->background_mapper()->tsdf_layer().numAllocatedBlocks();
    if (human_pointcloud_C_device_.size() != 60 * 40 || human_blocks < 5 || static_blocks < 50 || multi_mapper_->getLastDepthFrameMaskOverlay().rows() != 120) {
      std::fprintf(stderr, "human mapping: %d human points, %d human blocks, %d static blocks\n", human_pointcloud_C_device_.size(), human_blocks, static_blocks); return 1; }
  }
  // mesh streaming under a bandwidth limit (layer_streamer_bandwidth_limit_mbps, layer_publishing.cpp:702-711): a full-layer
  // mesh update is rationed over several calls, every block arrives exactly once
  {
    node.static_mapper_->updateColorMesh(UpdateFullLayer::kYes);
    node.static_mapper_->serializeSelectedLayers(LayerType::kColorMesh);                // no limit: the whole update at once
    const size_t full_blocks = node.static_mapper_->serializedColorMeshLayer()->block_indices.size();
    node.static_mapper_->updateColorMesh(UpdateFullLayer::kYes);
    size_t total_blocks = 0, calls = 0, first_call_blocks = 0;
    do {
      node.static_mapper_->serializeSelectedLayers(LayerType::kColorMesh, 0.8f);        // 0.8 Mbit/s x <= 1 s = <= 100 kB per call
      const size_t nb = node.static_mapper_->serializedColorMeshLayer()->block_indices.size();
      if (calls == 0) first_call_blocks = nb;
      total_blocks += nb; calls++;
    } while (node.static_mapper_->numMeshBlocksAwaitingStreaming() > 0 && calls < 100000);
    if (total_blocks != full_blocks || calls < 2 || first_call_blocks == 0 || first_call_blocks >= full_blocks) {
      std::fprintf(stderr, "rationed mesh streaming: %zu of %zu blocks in %zu calls (first call %zu)\n", total_blocks, full_blocks, calls, first_call_blocks); return 1; }
    // camera batches through the facade: the same two frames as ONE launch set == two calls (block sets and voxel sums)
    {
      DepthImage d0(MemoryType::kDevice), d1(MemoryType::kDevice); ColorImage c0(MemoryType::kDevice), c1(MemoryType::kDevice);
      std::vector<float> hd((size_t)rows * cols, 2.0f), hd2((size_t)rows * cols, 2.6f); std::vector<Color> hc((size_t)rows * cols, Color(10, 200, 90));
      d0.copyFromAsync(rows, cols, hd.data(), *node.cuda_stream_); d1.copyFromAsync(rows, cols, hd2.data(), *node.cuda_stream_);
      c0.copyFromAsync(rows, cols, hc.data(), *node.cuda_stream_); c1.copyFromAsync(rows, cols, hc.data(), *node.cuda_stream_);
      node.cuda_stream_->synchronize();
      Transform Ta = Transform::Identity(), Tb = Transform::Identity(); Tb.setTranslation(Vector3f(0.3f, 0.1f, 0.0f));
      Mapper batched(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, node.cuda_stream_), separate(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, node.cuda_stream_);
      batched.integrateDepthBatch({&d0, &d1}, {Ta, Tb}, {camera, camera}); batched.integrateColorBatch({&c0, &c1}, {Ta, Tb}, {camera, camera});
      separate.integrateDepth(d0, Ta, camera); separate.integrateDepth(d1, Tb, camera); separate.integrateColor(c0, Ta, camera); separate.integrateColor(c1, Tb, camera);
      double sa = 0.0, sb = 0.0; size_t na = 0, nb2 = 0;
      callFunctionOnAllVoxels<TsdfVoxel>(batched.tsdf_layer(), [&](const Index3D&, const Index3D&, const TsdfVoxel* v) { if (v->weight > 0.f) { sa += (double)v->distance * v->weight; na++; } });
      callFunctionOnAllVoxels<TsdfVoxel>(separate.tsdf_layer(), [&](const Index3D&, const Index3D&, const TsdfVoxel* v) { if (v->weight > 0.f) { sb += (double)v->distance * v->weight; nb2++; } });
      if (na != nb2 || na < 1000 || sa != sb || batched.tsdf_layer().numAllocatedBlocks() != separate.tsdf_layer().numAllocatedBlocks() ||
          batched.color_layer().numAllocatedBlocks() != separate.color_layer().numAllocatedBlocks()) {
        std::fprintf(stderr, "camera batch: %zu vs %zu observed voxels, sums %.9g vs %.9g\n", na, nb2, sa, sb); return 1; }
    }
    // voxel-layer streams under the same limit (layer_publishing.cpp:702-711): nearest blocks first, cut at the budget
    {
      BlockExclusionParams ex; ex.exclusion_center_m = Vector3f(0.5f, 0.25f, 1.0f); ex.exclusion_height_m = -1.0f; ex.exclusion_radius_m = -1.0f;
      node.static_mapper_->serializeSelectedLayers(LayerType::kTsdf | LayerType::kColor, -1.0f, ex);
      const std::vector<Index3D> all = node.static_mapper_->serializedTsdfLayer()->block_indices;
      node.static_mapper_->serializeSelectedLayers(LayerType::kTsdf | LayerType::kColor, 4.0f, ex);        // 4 Mbit/s x <= 1 s = 500 kB = 61 blocks of 8204 B
      const std::vector<Index3D> some = node.static_mapper_->serializedTsdfLayer()->block_indices;
      const float bs = node.static_mapper_->tsdf_layer().block_size();
      auto d2 = [&](const Index3D& b) { const Vector3f c = getCenterPositionFromBlockIndex(bs, b); const Vector3f d = c - ex.exclusion_center_m; return d.x() * d.x() + d.y() * d.y() + d.z() * d.z(); };
      float far_kept = 0.f, near_dropped = 1e30f;
      for (const Index3D& b : some) far_kept = std::max(far_kept, d2(b));
      for (const Index3D& b : all) if (!std::binary_search(some.begin(), some.end(), b)) near_dropped = std::min(near_dropped, d2(b));
      if (some.empty() || some.size() > 61 || some.size() >= all.size() || far_kept > near_dropped ||
          node.static_mapper_->serializedColorLayer()->block_indices.size() != some.size()) {
        std::fprintf(stderr, "rationed voxel-layer streaming: %zu of %zu blocks, farthest kept %g nearest dropped %g\n", some.size(), all.size(), far_kept, near_dropped); return 1; }
    }
  }
  // esdf_mode "3d" (node_params.hpp:90; nvblox_node.cpp:187-190): the ESDF of every voxel, sampled like the EsdfAndGradients service does
  {
    MultiMapper m3(0.05f, MappingType::kStaticTsdf, EsdfMode::k3D, MemoryType::kDevice, std::make_shared<CudaStreamOwning>(), 1 << 12);
    DepthImage flat(120, 160, MemoryType::kDevice);
    std::vector<float> host_depth(120 * 160, 1.5f);
    flat.copyFromAsync(120, 160, host_depth.data(), CudaStreamOwning());
    m3.integrateDepth(flat, Transform::Identity(), Camera(80.f, 80.f, 79.5f, 59.5f, 160, 120));
    m3.updateEsdf();
    const EsdfLayer& e3 = m3.background_mapper()->esdf_layer();
    size_t sites = 0, known = 0; int zmin = 1 << 30, zmax = -(1 << 30);
    callFunctionOnAllVoxels<EsdfVoxel>(e3, [&](const Index3D& b, const Index3D&, const EsdfVoxel* v) {
      sites += v->is_site ? 1 : 0; known += v->observed ? 1 : 0;
      zmin = std::min(zmin, b.z()); zmax = std::max(zmax, b.z()); });
    if (e3.numAllocatedBlocks() != m3.background_mapper()->tsdf_layer().numAllocatedBlocks() || sites < 1000 || known < 10000 || zmax - zmin < 3) {
      std::fprintf(stderr, "3-D ESDF: %d blocks, %zu sites, %zu observed, z %d..%d\n", e3.numAllocatedBlocks(), sites, known, zmin, zmax); return 1; }
  }
  // mapping_type "dynamic" (specializations/nvblox_dynamics.yaml): freespace layer, dynamic-pixel detection, mask clean-up, occupancy mapper
  {
    auto multi_mapper_ = std::make_shared<MultiMapper>(0.05f, MappingType::kDynamic, EsdfMode::k2D, MemoryType::kDevice, std::make_shared<CudaStreamOwning>(), 1 << 12);
    MapperParams sp; sp.freespace_integrator_params.min_duration_since_occupied_for_freespace_ms = Time(250);
    MapperParams dp; dp.occupancy_integrator_params.occupied_region_occupancy_probability = 0.9f; dp.occupancy_integrator_params.free_region_occupancy_probability = 0.2f;
    multi_mapper_->setMapperParams(sp, dp);
    MultiMapperParams mmp; mmp.connected_mask_component_size_threshold = 50;
    multi_mapper_->setMultiMapperParams(mmp);
    const Camera depth_camera_(80.f, 80.f, 79.5f, 59.5f, 160, 120);
    DepthImage depth_image_(120, 160, MemoryType::kDevice);
    std::vector<float> wall(120 * 160, 3.0f), with_object = wall;
    for (int r = 40; r < 90; r++) for (int c = 60; c < 100; c++) with_object[r * 160 + c] = 1.5f;          // something appears 1.5 m away, in mapped freespace
    int64_t t_ms = 0;
    for (int k = 0; k < 8; k++, t_ms += 100) {
      depth_image_.copyFromAsync(120, 160, wall.data(), CudaStreamOwning());
      multi_mapper_->integrateDepth(depth_image_, Transform::Identity(), depth_camera_, Time(t_ms));
    }
    depth_image_.copyFromAsync(120, 160, with_object.data(), CudaStreamOwning());
    multi_mapper_->integrateDepth(depth_image_, Transform::Identity(), depth_camera_, Time(t_ms));
    multi_mapper_->updateEsdf();
    const int dynamic_points = multi_mapper_->getLastDynamicPointcloud().size();
    size_t free_voxels = 0;
    callFunctionOnAllVoxels<FreespaceVoxel>(multi_mapper_->background_mapper()->freespace_layer(), [&](const Index3D&, const Index3D&, const FreespaceVoxel* v) { if (v->is_high_confidence_freespace) free_voxels++; });
    const int dyn_blocks = multi_mapper_->foreground_mapper()->occupancy_layer().numAllocatedBlocks();
    if (dynamic_points < 1500 || dynamic_points > 50 * 40 || free_voxels < 10000 || dyn_blocks < 3) {
      std::fprintf(stderr, "dynamic mapping: %d dynamic points, %zu freespace voxels, %d dynamic blocks\n", dynamic_points, free_voxels, dyn_blocks); return 1; }
    // the freespace and occupancy streams are cut to the exclusion cylinder and rationed like the TSDF stream (layer_publishing.cpp:702-711
    // hands the limit and the exclusion parameters to serializeSelectedLayers for every layer type)
    BlockExclusionParams ex; ex.exclusion_center_m = Vector3f(0.0f, 0.0f, 0.0f); ex.exclusion_height_m = -1.0f; ex.exclusion_radius_m = -1.0f;
    auto bg = multi_mapper_->background_mapper(); auto fg = multi_mapper_->foreground_mapper();
    bg->serializeSelectedLayers(LayerType::kFreespace, -1.0f, ex);
    const size_t fs_all = bg->serializedFreespaceLayer()->block_indices.size();
    bg->serializeSelectedLayers(LayerType::kFreespace, 2.0f, ex);           // 2 Mbit/s x <= 1 s = 250 kB = 30 blocks of 8204 B
    const std::vector<Index3D> fs_some = bg->serializedFreespaceLayer()->block_indices;
    const float bs = bg->tsdf_layer().block_size();
    float far_kept = 0.f;
    for (const Index3D& b : fs_some) { const Vector3f c = getCenterPositionFromBlockIndex(bs, b); far_kept = std::max(far_kept, c.x() * c.x() + c.y() * c.y() + c.z() * c.z()); }
    ex.exclusion_radius_m = 0.5f;                                               // a 0.5 m cylinder around the optical axis' origin
    fg->serializeSelectedLayers(LayerType::kOccupancy, -1.0f, ex);
    const size_t occ_in = fg->serializedOccupancyLayer()->block_indices.size();
    ex.exclusion_radius_m = -1.0f;
    fg->serializeSelectedLayers(LayerType::kOccupancy, 0.001f, ex);             // below one block's worth: exactly one block goes out
    if (fs_all < 100 || fs_some.empty() || fs_some.size() > 30 || far_kept > 1.6f * 1.6f || occ_in < 1 || occ_in > (size_t)dyn_blocks ||
        fg->serializedOccupancyLayer()->block_indices.size() != 1) {
      std::fprintf(stderr, "rationed freespace / occupancy streams: freespace %zu of %zu (farthest %g), occupancy %zu in the cylinder of %d, %zu under a tiny budget\n",
                   fs_some.size(), fs_all, far_kept, occ_in, dyn_blocks, fg->serializedOccupancyLayer()->block_indices.size()); return 1; }
  }
  // save_map / load_map services (nvblox_node.cpp:1668, 1703): bool results, a missing file is a recoverable error
  const std::string filename = std::string(argv[1]) + ".map";
  const bool save_ok = node.static_mapper_->saveLayerCake(filename);
  const int blocks_before = node.static_mapper_->tsdf_layer().numAllocatedBlocks();
  const bool load_missing = node.static_mapper_->loadMap(filename + ".does_not_exist");
  const bool load_ok = node.static_mapper_->loadMap(filename);
  if (!save_ok || load_missing || !load_ok || node.static_mapper_->tsdf_layer().numAllocatedBlocks() != blocks_before) {
    std::fprintf(stderr, "map save/load failed: save %d load_missing %d load %d\n", (int)save_ok, (int)load_missing, (int)load_ok); return 1; }
  std::printf("{\"tsdf_blocks\": %d, \"color_blocks\": %d, \"esdf_blocks\": %d, \"tsdf_observed\": %zu, \"tsdf_sum\": %.9g, "
              "\"slice_width\": %d, \"slice_height\": %d, \"slice_known\": %zu, \"slice_sum\": %.9g, \"occupied\": %zu, "
              "\"serialized_blocks\": %zu, \"serialized_visible\": %zu, \"aabb_min\": [%.6f, %.6f, %.6f], \"mesh_blocks\": %zu, \"mesh_vertices\": %zu, \"mesh_triangle_indices\": %zu, \"mesh_vertex_sum\": %.9g}\n",
              tsdf.numAllocatedBlocks(), node.static_mapper_->color_layer().numAllocatedBlocks(), node.static_mapper_->esdf_layer().numAllocatedBlocks(),
              observed, tsdf_sum, width, height, known, slice_sum, occupied, st->block_indices.size(), ser_visible, aabb.min().x(), aabb.min().y(), aabb.min().z(), mb, mv, mt, vsum);
  return 0;
}
