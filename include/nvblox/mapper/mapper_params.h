// nvblox/mapper/mapper_params.h -- MapperParams with the member names nvblox_ros writes
// (nvblox_ros/src/lib/mapper_initialization.cpp:231-466).  Members outside the static TSDF / colour / ESDF-2D / mesh
// path are carried (so getMapperParamsFromROS compiles) but ignored by libnvblox_hip; toCAbi() lists what is consumed.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "nvblox/core/parameter_tree.h"
#include "nvblox/core/types.h"
#include "nvblox/integrators/weighting_function.h"
#include "nvblox/utils/params.h"
#include "nvblox_hip.h"

namespace nvblox {

enum class WorkspaceBoundsType { kUnbounded = 0, kHeightBounds = 1, kBoundingBox = 2 };   // mapper_initialization.cpp:62-80
enum class EsdfMode { k3D, k2D };                                             // node_params.hpp:86-91
enum class MappingType { kStaticTsdf, kStaticOccupancy, kDynamic, kHumanWithStaticTsdf, kHumanWithStaticOccupancy };
enum class ProjectiveLayerType { kTsdf, kOccupancy, kTsdfWithFreespace, kNone };     // (layer_publishing.cpp:747 uses kTsdfWithFreespace)
enum class UpdateFullLayer { kNo, kYes };


// Parameter descriptions the ROS layer declares and reads through (mapper_initialization.cpp:113-228, 231-466).  Names are
// the ROS parameter names of nvblox_base.yaml:66-110; defaults are the values shipped there (the core's own defaults are
// not visible in the reference tree).
constexpr Param<bool>::Description kDoDepthPrepocessingParamDesc{"do_depth_preprocessing", false, "Dilate invalid-depth regions before integration."};
constexpr Param<int>::Description kDepthPreprocessingNumDilationsParamDesc{"depth_preprocessing_num_dilations", 3, "Pixels by which invalid-depth regions grow."};
constexpr Param<float>::Description kEsdfSliceMinHeightParamDesc{"esdf_slice_min_height", 0.0f, "Lower bound of the TSDF z band of the 2-D ESDF slice [m]."};
constexpr Param<float>::Description kEsdfSliceMaxHeightParamDesc{"esdf_slice_max_height", 1.0f, "Upper bound of the TSDF z band of the 2-D ESDF slice [m]."};
constexpr Param<float>::Description kEsdfSliceHeightParamDesc{"esdf_slice_height", 1.0f, "Output height of the 2-D ESDF slice [m]."};
constexpr Param<float>::Description kSliceHeightAbovePlaneMParamDesc{"slice_height_above_plane_m", 0.0f, "Slice height above the ground plane [m] (ground-plane mode)."};
constexpr Param<float>::Description kSliceHeightThicknessMParamDesc{"slice_height_thickness_m", 0.0f, "Slice thickness [m] (ground-plane mode)."};
constexpr Param<float>::Description kProjectiveIntegratorMaxIntegrationDistanceMParamDesc{"projective_integrator_max_integration_distance_m", 5.0f, "Maximum camera integration distance [m]."};
constexpr Param<float>::Description kLidarProjectiveIntegratorMaxIntegrationDistanceMParamDesc{"lidar_projective_integrator_max_integration_distance_m", 10.0f, "Maximum LiDAR integration distance [m]."};
constexpr Param<float>::Description kProjectiveIntegratorTruncationDistanceVoxParamDesc{"projective_integrator_truncation_distance_vox", 4.0f, "TSDF truncation distance [voxels]."};
constexpr Param<WeightingFunctionType>::Description kProjectiveIntegratorWeightingModeParamDesc{"projective_integrator_weighting_mode", WeightingFunctionType::kInverseSquareTsdfDistancePenalty, "Measurement weighting function."};
constexpr Param<float>::Description kProjectiveIntegratorMaxWeightParamDesc{"projective_integrator_max_weight", 5.0f, "Weight clamp."};
constexpr Param<float>::Description kProjectiveTsdfIntegratorInvalidDepthDecayFactor{"projective_tsdf_integrator_invalid_depth_decay_factor", -1.0f, "Weight decay of voxels projecting onto invalid depth (< 0: off)."};
constexpr Param<float>::Description kFreeRegionOccupancyProbabilityParamDesc{"free_region_occupancy_probability", 0.45f, "Occupancy integrator (not provided by libnvblox_hip)."};
constexpr Param<float>::Description kOccupiedRegionOccupancyProbabilityParamDesc{"occupied_region_occupancy_probability", 0.55f, "Occupancy integrator (not provided by libnvblox_hip)."};
constexpr Param<float>::Description kUnobservedRegionOccupancyProbabilityParamDesc{"unobserved_region_occupancy_probability", 0.5f, "Occupancy integrator (not provided by libnvblox_hip)."};
constexpr Param<float>::Description kOccupiedRegionHalfWidthMParamDesc{"occupied_region_half_width_m", 0.1f, "Occupancy integrator (not provided by libnvblox_hip)."};
constexpr Param<int>::Description kRaycastSubsamplingFactorParamDesc{"raycast_subsampling_factor", 4, "Depth image sub-sampling of the view calculation rays."};
constexpr Param<WorkspaceBoundsType>::Description kWorkspaceBoundsTypeParamDesc{"workspace_bounds_type", WorkspaceBoundsType::kUnbounded, "unbounded / height_bounds / bounding_box."};
constexpr Param<float>::Description kWorkspaceBoundsMinHeightParamDesc{"workspace_bounds_min_height_m", -0.5f, "Workspace bounds [m]."};
constexpr Param<float>::Description kWorkspaceBoundsMaxHeightParamDesc{"workspace_bounds_max_height_m", 2.0f, "Workspace bounds [m]."};
constexpr Param<float>::Description kWorkspaceBoundsMinCornerXParamDesc{"workspace_bounds_min_corner_x_m", 0.0f, "Workspace bounds [m]."};
constexpr Param<float>::Description kWorkspaceBoundsMaxCornerXParamDesc{"workspace_bounds_max_corner_x_m", 0.0f, "Workspace bounds [m]."};
constexpr Param<float>::Description kWorkspaceBoundsMinCornerYParamDesc{"workspace_bounds_min_corner_y_m", 0.0f, "Workspace bounds [m]."};
constexpr Param<float>::Description kWorkspaceBoundsMaxCornerYParamDesc{"workspace_bounds_max_corner_y_m", 0.0f, "Workspace bounds [m]."};
constexpr Param<float>::Description kEsdfIntegratorMinWeightParamDesc{"esdf_integrator_min_weight", 0.1f, "Minimum TSDF weight for a voxel to count as observed."};
constexpr Param<float>::Description kEsdfIntegratorMaxSiteDistanceVoxParamDesc{"esdf_integrator_max_site_distance_vox", 2.0f, "Maximum |TSDF| of a site [voxels]."};
constexpr Param<float>::Description kEsdfIntegratorMaxDistanceMParamDesc{"esdf_integrator_max_distance_m", 2.0f, "ESDF cut-off distance [m]."};
constexpr Param<float>::Description kMeshIntegratorMinWeightParamDesc{"mesh_integrator_min_weight", 0.1f, "Minimum TSDF weight of a meshed corner."};
constexpr Param<bool>::Description kMeshIntegratorWeldVerticesParamDesc{"mesh_integrator_weld_vertices", true, "Weld vertices per block."};
constexpr Param<bool>::Description kDecayIntegratorDeallocateDecayedBlocks{"decay_integrator_deallocate_decayed_blocks", true, "Deallocate fully decayed blocks (false: they stay allocated with their decayed voxels)."};
constexpr Param<float>::Description kTsdfDecayFactorParamDesc{"tsdf_decay_factor", 0.95f, "Weight multiplier per decay step."};
constexpr Param<float>::Description kTsdfDecayedWeightThresholdDesc{"tsdf_decayed_weight_threshold", 0.001f, "Blocks whose weights are all below this are deallocated."};
constexpr Param<bool>::Description kTsdfSetFreeDistanceOnDecayedDesc{"tsdf_set_free_distance_on_decayed", false, "An observed voxel whose weight decays below the threshold becomes free (distance = tsdf_decayed_free_distance_vox, weight = the threshold) instead of unknown."};
constexpr Param<float>::Description kTsdfDecayedFreeDistanceVoxDesc{"tsdf_decayed_free_distance_vox", 4.0f, "Distance (voxels) given to a decayed voxel when tsdf_set_free_distance_on_decayed is set."};
constexpr Param<float>::Description kFreeRegionDecayProbabilityParamDesc{"free_region_decay_probability", 0.55f, "Decay probability applied to free voxels (log-odds step towards unknown)."};
constexpr Param<float>::Description kOccupiedRegionDecayProbabilityParamDesc{"occupied_region_decay_probability", 0.4f, "Decay probability applied to occupied voxels (log-odds step towards unknown)."};
constexpr Param<bool>::Description kOccupancyDecayToFreeParamDesc{"occupancy_decay_to_free", false, "Occupied voxels decay past unknown into free and stay there; free voxels are not decayed."};
constexpr Param<float>::Description kMaxTsdfDistanceForOccupancyMParamDesc{"max_tsdf_distance_for_occupancy_m", 0.15f, "Freespace integrator (dynamic mapping)."};
constexpr Param<int>::Description kMaxUnobservedToKeepConsecutiveOccupancyMsParamDesc{"max_unobserved_to_keep_consecutive_occupancy_ms", 200, "Freespace integrator (dynamic mapping)."};
constexpr Param<int>::Description kMinDurationSinceOccupiedForFreespaceMsParamDesc{"min_duration_since_occupied_for_freespace_ms", 1000, "Freespace integrator (dynamic mapping)."};
constexpr Param<int>::Description kMinConsecutiveOccupancyDurationForResetMsParamDesc{"min_consecutive_occupancy_duration_for_reset_ms", 2000, "Freespace integrator (dynamic mapping)."};
constexpr Param<bool>::Description kCheckNeighborhoodParamDesc{"check_neighborhood", true, "Freespace integrator (dynamic mapping)."};
constexpr Param<bool>::Description kInitializeToHighConfidenceFreespaceParamDesc{"initialize_to_high_confidence_freespace", false, "Freespace integrator (dynamic mapping)."};
constexpr Param<int>::Description kConnectedMaskComponentSizeThresholdParamDesc{"connected_mask_component_size_threshold", 2000, "MultiMapper (dynamic / human mapping): mask clean-up."};
constexpr Param<bool>::Description kRemoveSmallConnectedComponentsParamDesc{"remove_small_connected_components", true, "MultiMapper (dynamic / human mapping): mask clean-up."};

struct ProjectiveIntegratorParams {
  float projective_integrator_max_integration_distance_m = 7.0f;
  float lidar_projective_integrator_max_integration_distance_m = 10.0f;
  float projective_integrator_truncation_distance_vox = 4.0f;
  WeightingFunctionType projective_integrator_weighting_mode = WeightingFunctionType::kInverseSquareWeight;
  float projective_integrator_max_weight = 5.0f;
  float projective_tsdf_integrator_invalid_depth_decay_factor = -1.0f;
};
struct ViewCalculatorParams {
  int raycast_subsampling_factor = 4;
  WorkspaceBoundsType workspace_bounds_type = WorkspaceBoundsType::kUnbounded;
  float workspace_bounds_min_height_m = 0.f, workspace_bounds_max_height_m = 0.f;
  float workspace_bounds_min_corner_x_m = 0.f, workspace_bounds_max_corner_x_m = 0.f;
  float workspace_bounds_min_corner_y_m = 0.f, workspace_bounds_max_corner_y_m = 0.f;
};
struct EsdfIntegratorParams {
  float esdf_integrator_min_weight = 0.1f;
  float esdf_integrator_max_site_distance_vox = 2.0f;
  float esdf_integrator_max_distance_m = 2.0f;
  float esdf_slice_min_height = 0.0f, esdf_slice_max_height = 1.0f, esdf_slice_height = 1.0f;
  float slice_height_above_plane_m = 0.0f, slice_height_thickness_m = 0.0f;
};
struct MeshIntegratorParams { float mesh_integrator_min_weight = 0.1f; bool mesh_integrator_weld_vertices = true; };
struct DecayIntegratorBaseParams { bool decay_integrator_deallocate_decayed_blocks = true; };
struct TsdfDecayIntegratorParams {
  float tsdf_decay_factor = 0.95f, tsdf_decayed_weight_threshold = 0.001f;
  bool tsdf_set_free_distance_on_decayed = false; float tsdf_decayed_free_distance_vox = 4.0f;
};
struct OccupancyIntegratorParams {
  float free_region_occupancy_probability = 0.3f, occupied_region_occupancy_probability = 0.7f,
        unobserved_region_occupancy_probability = 0.5f, occupied_region_half_width_m = 0.1f;
};
struct OccupancyDecayIntegratorParams { float free_region_decay_probability = 0.55f, occupied_region_decay_probability = 0.4f; bool occupancy_decay_to_free = false; };
struct FreespaceIntegratorParams {
  float max_tsdf_distance_for_occupancy_m = 0.15f; Time max_unobserved_to_keep_consecutive_occupancy_ms{200};
  Time min_duration_since_occupied_for_freespace_ms{1000}; Time min_consecutive_occupancy_duration_for_reset_ms{2000};
  bool check_neighborhood = true, initialize_to_high_confidence_freespace = false;
};

struct MapperParams {
  bool do_depth_preprocessing = false;
  int depth_preprocessing_num_dilations = 4;
  ProjectiveIntegratorParams projective_integrator_params;
  ViewCalculatorParams view_calculator_params;
  EsdfIntegratorParams esdf_integrator_params;
  MeshIntegratorParams mesh_integrator_params;
  DecayIntegratorBaseParams decay_integrator_base_params;
  TsdfDecayIntegratorParams tsdf_decay_integrator_params;
  OccupancyIntegratorParams occupancy_integrator_params;
  OccupancyDecayIntegratorParams occupancy_decay_integrator_params;
  FreespaceIntegratorParams freespace_integrator_params;

  // Mapper::getParameterTree (nvblox_node.cpp:120): every knob by the name mapper_initialization.cpp declares it under
  parameters::ParameterTreeNode getParameterTree(const std::string& name = "mapper") const {
    using N = parameters::ParameterTreeNode;
    const auto& pi = projective_integrator_params; const auto& vc = view_calculator_params; const auto& es = esdf_integrator_params;
    const auto& fs = freespace_integrator_params;
    return N(name, std::vector<N>{
        N("do_depth_preprocessing", do_depth_preprocessing), N("depth_preprocessing_num_dilations", depth_preprocessing_num_dilations),
        N("projective_integrator", std::vector<N>{
            N("projective_integrator_max_integration_distance_m", pi.projective_integrator_max_integration_distance_m),
            N("lidar_projective_integrator_max_integration_distance_m", pi.lidar_projective_integrator_max_integration_distance_m),
            N("projective_integrator_truncation_distance_vox", pi.projective_integrator_truncation_distance_vox),
            N("projective_integrator_weighting_mode", pi.projective_integrator_weighting_mode),
            N("projective_integrator_max_weight", pi.projective_integrator_max_weight),
            N("projective_tsdf_integrator_invalid_depth_decay_factor", pi.projective_tsdf_integrator_invalid_depth_decay_factor)}),
        N("view_calculator", std::vector<N>{
            N("raycast_subsampling_factor", vc.raycast_subsampling_factor), N("workspace_bounds_type", vc.workspace_bounds_type),
            N("workspace_bounds_min_height_m", vc.workspace_bounds_min_height_m), N("workspace_bounds_max_height_m", vc.workspace_bounds_max_height_m),
            N("workspace_bounds_min_corner_x_m", vc.workspace_bounds_min_corner_x_m), N("workspace_bounds_max_corner_x_m", vc.workspace_bounds_max_corner_x_m),
            N("workspace_bounds_min_corner_y_m", vc.workspace_bounds_min_corner_y_m), N("workspace_bounds_max_corner_y_m", vc.workspace_bounds_max_corner_y_m)}),
        N("esdf_integrator", std::vector<N>{
            N("esdf_integrator_min_weight", es.esdf_integrator_min_weight), N("esdf_integrator_max_site_distance_vox", es.esdf_integrator_max_site_distance_vox),
            N("esdf_integrator_max_distance_m", es.esdf_integrator_max_distance_m), N("esdf_slice_min_height", es.esdf_slice_min_height),
            N("esdf_slice_max_height", es.esdf_slice_max_height), N("esdf_slice_height", es.esdf_slice_height)}),
        N("mesh_integrator", std::vector<N>{
            N("mesh_integrator_min_weight", mesh_integrator_params.mesh_integrator_min_weight),
            N("mesh_integrator_weld_vertices", mesh_integrator_params.mesh_integrator_weld_vertices)}),
        N("decay_integrators", std::vector<N>{
            N("decay_integrator_deallocate_decayed_blocks", decay_integrator_base_params.decay_integrator_deallocate_decayed_blocks),
            N("tsdf_decay_factor", tsdf_decay_integrator_params.tsdf_decay_factor),
            N("tsdf_decayed_weight_threshold", tsdf_decay_integrator_params.tsdf_decayed_weight_threshold),
            N("free_region_decay_probability", occupancy_decay_integrator_params.free_region_decay_probability),
            N("occupied_region_decay_probability", occupancy_decay_integrator_params.occupied_region_decay_probability)}),
        N("occupancy_integrator", std::vector<N>{
            N("free_region_occupancy_probability", occupancy_integrator_params.free_region_occupancy_probability),
            N("occupied_region_occupancy_probability", occupancy_integrator_params.occupied_region_occupancy_probability),
            N("unobserved_region_occupancy_probability", occupancy_integrator_params.unobserved_region_occupancy_probability),
            N("occupied_region_half_width_m", occupancy_integrator_params.occupied_region_half_width_m)}),
        N("freespace_integrator", std::vector<N>{
            N("max_tsdf_distance_for_occupancy_m", fs.max_tsdf_distance_for_occupancy_m),
            N("max_unobserved_to_keep_consecutive_occupancy_ms", static_cast<int64_t>(fs.max_unobserved_to_keep_consecutive_occupancy_ms)),
            N("min_duration_since_occupied_for_freespace_ms", static_cast<int64_t>(fs.min_duration_since_occupied_for_freespace_ms)),
            N("min_consecutive_occupancy_duration_for_reset_ms", static_cast<int64_t>(fs.min_consecutive_occupancy_duration_for_reset_ms)),
            N("check_neighborhood", fs.check_neighborhood), N("initialize_to_high_confidence_freespace", fs.initialize_to_high_confidence_freespace)})});
  }

  // [U] open choices of libnvblox_hip (include/nvblox_hip.h): not reference parameters -- the switches that pin this
  // implementation to the real nvblox core once its source can be read; defaults = the documented behaviour
  struct HipOpenChoices {
    int tsdf_weighting_variant = 0, tsdf_skip_at_negative_truncation = 0, tsdf_weight_clamp_before_blend = 0;
    float color_occlusion_threshold_vox = -1.0f;
    int esdf_propagation = 0, mesh_ambiguity_rule = 0, mesh_normal_rule = 0, esdf_site_rule = 0, depth_interp_nearest = 0;
  } hip_open_choices;

  // what libnvblox_hip consumes
  nvbx_mapper_params toCAbi(float voxel_size, ProjectiveLayerType layer_type = ProjectiveLayerType::kTsdf, EsdfMode esdf_mode = EsdfMode::k2D) const {
    nvbx_mapper_params p;
    nvbx_default_params(&p);      // every field this function does not set keeps the library default (incl. the [U] open-choice switches)
    // the decay integrators' switches (mapper_initialization.cpp:383-428): every value the node can set is honoured
    p.decay_deallocate_decayed_blocks = decay_integrator_base_params.decay_integrator_deallocate_decayed_blocks ? 1 : 0;
    p.tsdf_set_free_distance_on_decayed = tsdf_decay_integrator_params.tsdf_set_free_distance_on_decayed ? 1 : 0;
    p.tsdf_decayed_free_distance_vox = tsdf_decay_integrator_params.tsdf_decayed_free_distance_vox;
    p.occupancy_decay_to_free = occupancy_decay_integrator_params.occupancy_decay_to_free ? 1 : 0;
    p.voxel_size = voxel_size;
    p.esdf_mode = esdf_mode == EsdfMode::k3D ? 1 : 0;
    p.projective_layer_type = layer_type == ProjectiveLayerType::kOccupancy ? 1 : (layer_type == ProjectiveLayerType::kTsdfWithFreespace ? 2 : 0);
    p.max_tsdf_distance_for_occupancy_m = freespace_integrator_params.max_tsdf_distance_for_occupancy_m;
    p.max_unobserved_to_keep_consecutive_occupancy_ms = (int32_t)(int64_t)freespace_integrator_params.max_unobserved_to_keep_consecutive_occupancy_ms;
    p.min_duration_since_occupied_for_freespace_ms = (int32_t)(int64_t)freespace_integrator_params.min_duration_since_occupied_for_freespace_ms;
    p.min_consecutive_occupancy_duration_for_reset_ms = (int32_t)(int64_t)freespace_integrator_params.min_consecutive_occupancy_duration_for_reset_ms;
    p.check_neighborhood = freespace_integrator_params.check_neighborhood ? 1 : 0;
    p.initialize_to_high_confidence_freespace = freespace_integrator_params.initialize_to_high_confidence_freespace ? 1 : 0;
    p.free_region_occupancy_probability = occupancy_integrator_params.free_region_occupancy_probability;
    p.occupied_region_occupancy_probability = occupancy_integrator_params.occupied_region_occupancy_probability;
    p.unobserved_region_occupancy_probability = occupancy_integrator_params.unobserved_region_occupancy_probability;
    p.occupied_region_half_width_m = occupancy_integrator_params.occupied_region_half_width_m;
    p.free_region_decay_probability = occupancy_decay_integrator_params.free_region_decay_probability;
    p.occupied_region_decay_probability = occupancy_decay_integrator_params.occupied_region_decay_probability;
    p.max_integration_distance_m = projective_integrator_params.projective_integrator_max_integration_distance_m;
    p.truncation_distance_vox = projective_integrator_params.projective_integrator_truncation_distance_vox;
    p.max_weight = projective_integrator_params.projective_integrator_max_weight;
    p.weighting_mode = (int32_t)projective_integrator_params.projective_integrator_weighting_mode;
    p.raycast_subsampling_factor = view_calculator_params.raycast_subsampling_factor;
    p.esdf_min_weight = esdf_integrator_params.esdf_integrator_min_weight;
    p.esdf_max_site_distance_vox = esdf_integrator_params.esdf_integrator_max_site_distance_vox;
    p.esdf_max_distance_m = esdf_integrator_params.esdf_integrator_max_distance_m;
    p.esdf_slice_height = esdf_integrator_params.esdf_slice_height;
    p.esdf_slice_min_height = esdf_integrator_params.esdf_slice_min_height;
    p.esdf_slice_max_height = esdf_integrator_params.esdf_slice_max_height;
    p.slice_height_above_plane_m = esdf_integrator_params.slice_height_above_plane_m;      // (mapper_initialization.cpp:257-260; used once a ground plane is set: Mapper::setEsdfGroundPlane)
    p.slice_height_thickness_m = esdf_integrator_params.slice_height_thickness_m;
    p.mesh_min_weight = mesh_integrator_params.mesh_integrator_min_weight;
    p.mesh_weld_vertices = mesh_integrator_params.mesh_integrator_weld_vertices ? 1 : 0;
    p.sphere_tracing_subsampling = 4; p.sphere_tracing_max_steps = 100;
    p.sphere_tracing_max_ray_length_m = 15.0f; p.sphere_tracing_surface_eps_vox = 0.1f;
    p.tsdf_decay_factor = tsdf_decay_integrator_params.tsdf_decay_factor;
    p.tsdf_decayed_weight_threshold = tsdf_decay_integrator_params.tsdf_decayed_weight_threshold;
    p.lidar_max_integration_distance_m = projective_integrator_params.lidar_projective_integrator_max_integration_distance_m;
    p.lidar_linear_interpolation_max_allowable_difference_vox = 2.0f;
    p.lidar_nearest_interpolation_max_allowable_dist_to_ray_vox = 0.5f;
    p.invalid_depth_decay_factor = projective_integrator_params.projective_tsdf_integrator_invalid_depth_decay_factor;
    p.do_depth_preprocessing = do_depth_preprocessing ? 1 : 0; p.depth_preprocessing_num_dilations = depth_preprocessing_num_dilations;
    p.workspace_bounds_type = (int32_t)view_calculator_params.workspace_bounds_type;
    p.workspace_bounds_min_corner_m[0] = view_calculator_params.workspace_bounds_min_corner_x_m; p.workspace_bounds_min_corner_m[1] = view_calculator_params.workspace_bounds_min_corner_y_m;
    p.workspace_bounds_min_corner_m[2] = view_calculator_params.workspace_bounds_min_height_m;
    p.workspace_bounds_max_corner_m[0] = view_calculator_params.workspace_bounds_max_corner_x_m; p.workspace_bounds_max_corner_m[1] = view_calculator_params.workspace_bounds_max_corner_y_m;
    p.workspace_bounds_max_corner_m[2] = view_calculator_params.workspace_bounds_max_height_m;
    p.tsdf_weighting_variant = hip_open_choices.tsdf_weighting_variant; p.tsdf_skip_at_negative_truncation = hip_open_choices.tsdf_skip_at_negative_truncation;
    p.tsdf_weight_clamp_before_blend = hip_open_choices.tsdf_weight_clamp_before_blend; p.color_occlusion_threshold_vox = hip_open_choices.color_occlusion_threshold_vox;
    p.esdf_propagation = hip_open_choices.esdf_propagation; p.mesh_ambiguity_rule = hip_open_choices.mesh_ambiguity_rule; p.mesh_normal_rule = hip_open_choices.mesh_normal_rule;
    p.esdf_site_rule = hip_open_choices.esdf_site_rule; p.depth_interp_nearest = hip_open_choices.depth_interp_nearest;
    return p;
  }
};

struct MultiMapperParams {   // multi_mapper.* parameters (mapper_initialization.cpp: getMultiMapperParamsFromROS)
  int connected_mask_component_size_threshold = 2000;   // pixels (mapper_initialization.cpp:130)
  int remove_small_connected_components = 1;
  float mask_occlusion_threshold_m = 0.25f;     // [U] ImageMasker occlusion test (nvbx_split_depth_by_mask)
  // ground plane estimation (mapper_initialization.cpp:133-153; off in every shipped configuration)
  bool experimental_use_ground_plane_estimation = false;
  struct GroundPlaneEstimatorParams { float ground_points_candidates_min_z_m = -0.1f, ground_points_candidates_max_z_m = 0.15f; } ground_plane_estimator_params;
  struct RansacPlaneFitterParams { float ransac_distance_threshold_m = 0.15f; int num_ransac_iterations = 1000; } ransac_plane_fitter_params;
};

}  // namespace nvblox
