// fake_fuser_node.cpp -- FuserNode without ROS: the constructor's dataset switch and fuseNextFrame() of
// /root/reference/nvblox_ros/src/lib/fuser_node.cpp (:44-65 createFuser per dataset type, :85-97 setMultiMapper / setMapperParams /
// static_mapper, :202-224 the integrate loop, :229-232 the per-step getters, :258-273 ESDF slice, :276-287 serialized mesh,
// :290-304 back projection, :306-310 layer serialization), call expressions kept as in the reference, ROS publishers replaced by
// counters.  Compiled with g++ against include/nvblox/** (tests/cpp/Makefile) -- the boundary check for
// nvblox/executables/fuser.h and nvblox/datasets/{3dmatch,redwood,replica,data_loader_interface}.h.
//
// usage: fake_fuser_node <3dmatch|redwood|replica> <dataset_path> [number_of_frames_to_integrate]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <nvblox/nvblox.h>
#include "nvblox/datasets/3dmatch.h"
#include "nvblox/datasets/redwood.h"
#include "nvblox/datasets/replica.h"
#include "nvblox/integrators/esdf_slicer.h"

using namespace nvblox;

enum class RosDatasetType { kThreedMatch, kRedwood, kReplica };

struct FakeFuserNode {
  std::unique_ptr<CameraFuser> fuser_;
  std::shared_ptr<Mapper> mapper_;
  std::shared_ptr<CudaStream> cuda_stream_;
  EsdfSlicer esdf_slicer_;
  DepthImageBackProjector image_back_projector_;
  Pointcloud pointcloud_C_device_{MemoryType::kDevice}, pointcloud_L_device_{MemoryType::kDevice};
  int number_of_frames_to_integrate = -1, current_frame_number_ = 0;
  // what the publishers would have sent
  size_t frames = 0, bad_frames = 0, mesh_blocks_sent = 0, mesh_vertices_sent = 0, slice_pixels = 0, back_projected_points = 0, color_frames = 0, serialized_tsdf_blocks = 0;
  double depth_sum = 0.0;

  bool init(RosDatasetType dataset_type, const std::string& dataset_path) {
    cuda_stream_ = CudaStream::createCudaStream(CudaStreamType::kNonBlocking);
    constexpr int kSeqId = 1;
    constexpr bool kInitFromGflags = false;
    switch (dataset_type) {                                       // fuser_node.cpp:48-58
      case RosDatasetType::kThreedMatch: fuser_ = datasets::threedmatch::createFuser(dataset_path, kSeqId, kInitFromGflags); break;
      case RosDatasetType::kRedwood: fuser_ = datasets::redwood::createFuser(dataset_path, kInitFromGflags); break;
      case RosDatasetType::kReplica: fuser_ = datasets::replica::createFuser(dataset_path, kInitFromGflags); break;
    }
    if (!fuser_) return false;                                   // :68-74
    fuser_->setMultiMapper(std::make_shared<MultiMapper>(0.05f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, cuda_stream_));   // :85-89
    MapperParams mapper_params;                                  // fuser.yaml:24-42
    mapper_params.projective_integrator_params.projective_integrator_max_integration_distance_m = 8.0f;
    mapper_params.projective_integrator_params.projective_integrator_weighting_mode = WeightingFunctionType::kConstantWeight;
    mapper_params.esdf_integrator_params.esdf_slice_height = 0.09f; mapper_params.esdf_integrator_params.esdf_slice_min_height = 0.09f;
    mapper_params.esdf_integrator_params.esdf_slice_max_height = 0.65f;
    fuser_->multi_mapper()->setMapperParams(mapper_params);      // :94
    mapper_ = fuser_->static_mapper();                           // :97
    return mapper_ != nullptr;
  }

  bool fuseNextFrame() {                                          // fuser_node.cpp:202-313
    const bool fuse_next_frame = number_of_frames_to_integrate < 0 || current_frame_number_ < number_of_frames_to_integrate;
    if (!fuse_next_frame) return false;
    datasets::DataLoadResult fuse_result = fuser_->integrateFrame(current_frame_number_++);
    if (fuse_result == datasets::DataLoadResult::kBadFrame) { bad_frames++; return true; }
    else if (fuse_result == datasets::DataLoadResult::kNoMoreData) return false;
    const std::shared_ptr<const DepthImage> depth_frame = fuser_->getSensorData();
    const std::shared_ptr<const ColorImage> color_frame = fuser_->getColorFrame();
    const std::shared_ptr<const Camera> depth_camera = fuser_->getSensor();
    const std::shared_ptr<const Transform> depth_T_L_C = fuser_->getSensorPose();
    frames++;
    if (color_frame->numel() > 0) color_frames++;
    {   // "publish the depth frame": one D2H of the frame the loader produced
      std::vector<float> host((size_t)depth_frame->numel());
      depth_frame->copyToAsync(host.data(), *cuda_stream_); cuda_stream_->synchronize();
      for (float v : host) depth_sum += v;
    }
    {   // ESDF slice (:258-273)
      AxisAlignedBoundingBox aabb;
      Image<float> map_slice_image(MemoryType::kDevice);
      esdf_slicer_.sliceLayerToDistanceImage(mapper_->esdf_layer(), mapper_->esdf_integrator().esdf_slice_height(), 1000.0f, &aabb, &map_slice_image);
      slice_pixels = (size_t)map_slice_image.numel();
    }
    {   // mesh (:276-287)
      std::shared_ptr<SerializedColorMeshLayer> serialized_mesh = fuser_->getSerializedColorMesh();
      mesh_blocks_sent += serialized_mesh->block_indices.size();
      for (size_t b = 0; b < serialized_mesh->block_indices.size(); b++) mesh_vertices_sent += serialized_mesh->getNumVerticesInBlock(b);
    }
    {   // back projection (:290-304)
      image_back_projector_.backProjectOnGPU(*depth_frame, *depth_camera, &pointcloud_C_device_, 8.0f);
      transformPointcloudOnGPU(*depth_T_L_C, pointcloud_C_device_, &pointcloud_L_device_);
      back_projected_points = (size_t)pointcloud_L_device_.size();
    }
    {   // layers (:306-310 -> layer_publishing.cpp:702-711)
      BlockExclusionParams ex; ex.exclusion_center_m = depth_T_L_C->translation(); ex.exclusion_height_m = 2.0f; ex.exclusion_radius_m = 7.0f;
      mapper_->serializeSelectedLayers(LayerType::kTsdf | LayerType::kColor, -1.0f, ex);
      serialized_tsdf_blocks = mapper_->serializedTsdfLayer()->block_indices.size();
    }
    return true;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <3dmatch|redwood|replica> <dataset_path> [n_frames]\n", argv[0]); return 2; }
  const RosDatasetType type = !std::strcmp(argv[1], "3dmatch") ? RosDatasetType::kThreedMatch : (!std::strcmp(argv[1], "redwood") ? RosDatasetType::kRedwood : RosDatasetType::kReplica);
  warmupCuda();
  FakeFuserNode node;
  if (argc > 3) node.number_of_frames_to_integrate = std::atoi(argv[3]);
  if (!node.init(type, argv[2])) { std::fprintf(stderr, "Creation of %s fuser failed with dataset path: %s\n", argv[1], argv[2]); return 1; }
  while (node.fuseNextFrame()) {}
  const TsdfLayer& tsdf = node.mapper_->tsdf_layer();
  double tsdf_sum = 0.0; size_t observed = 0;
  callFunctionOnAllVoxels<TsdfVoxel>(tsdf, [&](const Index3D&, const Index3D&, const TsdfVoxel* v) { if (v->weight > 0.f) { tsdf_sum += (double)v->distance * (double)v->weight; observed++; } });
  std::printf("{\"frames\": %zu, \"bad_frames\": %zu, \"color_frames\": %zu, \"tsdf_blocks\": %d, \"color_blocks\": %d, \"esdf_blocks\": %d, \"tsdf_observed\": %zu, "
              "\"tsdf_sum\": %.9g, \"depth_sum\": %.9g, \"slice_pixels\": %zu, \"mesh_blocks_sent\": %zu, \"mesh_vertices_sent\": %zu, \"back_projected_points\": %zu, "
              "\"serialized_tsdf_blocks\": %zu}\n",
              node.frames, node.bad_frames, node.color_frames, tsdf.numAllocatedBlocks(), node.mapper_->color_layer().numAllocatedBlocks(),
              node.mapper_->esdf_layer().numAllocatedBlocks(), observed, tsdf_sum, node.depth_sum, node.slice_pixels, node.mesh_blocks_sent, node.mesh_vertices_sent,
              node.back_projected_points, node.serialized_tsdf_blocks);
  return 0;
}
