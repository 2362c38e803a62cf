/*
 * nvblox_hip.h -- C-ABI of libnvblox_hip.so, the MI355X (gfx950) implementation of the nvblox_core hot path
 * that isaac_ros_nvblox reaches through nvblox::MultiMapper / nvblox::Mapper / nvblox::EsdfSlicer.
 *
 * Every entry point names the reference call site it serves (paths relative to the reference root).  The C++ classes
 * in include/nvblox/ are inline wrappers over these functions, so nvblox_ros links against this library unchanged at
 * the call-expression level (INTEGRATION.md shows the CMake swap).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.  `*_dev` pointers are device (HBM) addresses on the mapper's
 *     GPU, everything else is host memory.
 *   - all work is enqueued on the mapper's stream (the reference is single-caller, stream-ordered:
 *     nvblox_ros/src/lib/nvblox_node.cpp:99,456-459).  Functions that return host-visible results synchronise that
 *     stream; the integrate / update calls do not.  (One piece of work may be enqueued one call late: see nvbx_update_esdf.)
 *   - return value: 0 = ok, negative = error (NVBX_E_*).  Nothing throws across the boundary; device faults are
 *     reported through nvbx_last_error().  (Reference convention: bool for I/O, CHECK/abort for programmer errors,
 *     checkCudaErrors -> exit(99): nvblox_ros_common/src/check_cuda_errors.cpp:24-32.)
 *   - transforms are row-major 4x4 float (T_L_C: camera -> layer/world), cameras are (fu,fv,cu,cv,width,height) as in
 *     conversions/image_conversions.cpp:27-32.
 *   - voxel blocks are 8x8x8, block copies use the reference's voxel structs and its linear order z + 8*y + 64*x
 *     (nvblox_ros/src/lib/layer_publishing.cpp:335,501).
 */
#ifndef NVBLOX_HIP_H_
#define NVBLOX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVBX_OK 0
#define NVBX_E_INVALID (-1)   /* bad argument */
#define NVBX_E_DEVICE (-2)    /* HIP runtime error, see nvbx_last_error() */
#define NVBX_E_CAPACITY (-3)  /* block pool / arena / window capacity exceeded */
#define NVBX_E_NOTFOUND (-4)
#define NVBX_E_IO (-5)        /* file could not be opened / read / written, or is not a map file */

/* layer selectors (bit mask), mirrors the reference's LayerType usage in layer_publishing.cpp:675-826 */
#define NVBX_LAYER_TSDF 1u
#define NVBX_LAYER_COLOR 2u
#define NVBX_LAYER_ESDF 4u
#define NVBX_LAYER_MESH 8u
#define NVBX_LAYER_OCCUPANCY 16u   /* occupancy mappers only; shares the projective voxel pool with TSDF */
#define NVBX_LAYER_FREESPACE 32u   /* projective_layer_type 2 only */

typedef struct nvbx_mapper nvbx_mapper; /* replaces nvblox::Mapper (one per GPU / stream) */

typedef struct { int32_t x, y, z; } nvbx_index3d;                      /* nvblox::Index3D; addressable range [-2^20, 2^20) per axis (419 km at 0.05 m voxels): poses / indices beyond it are refused with NVBX_E_INVALID */
typedef struct { float fu, fv, cu, cv; int32_t width, height; } nvbx_camera; /* nvblox::Camera */

/* Voxel structs as the reference's consumers read them. */
typedef struct { float distance, weight; } nvbx_tsdf_voxel;            /* layer_publishing.cpp:111,179 */
/* FreespaceVoxel: layer_publishing.cpp:129-137,158-165 (consecutive_occupancy_duration_ms, is_high_confidence_freespace) */
typedef struct { int64_t last_occupied_timestamp_ms; int64_t consecutive_occupancy_duration_ms; uint8_t is_high_confidence_freespace; uint8_t initialized; uint8_t pad[6]; } nvbx_freespace_voxel;
typedef struct { float log_odds; } nvbx_occupancy_voxel;                /* layer_publishing.cpp:140-154 (OccupancyVoxel::log_odds) */
typedef struct { uint8_t r, g, b, pad; float weight; } nvbx_color_voxel; /* layer_publishing.cpp:62-76; Color = 3 x u8 */
typedef struct {                                                        /* esdf_and_gradients_conversions.cu:33-44 */
  float squared_distance_vox; int32_t parent_direction[3]; uint8_t is_inside, observed, is_site, pad;
} nvbx_esdf_voxel;

/* WeightingFunctionType, nvblox_ros/src/lib/mapper_initialization.cpp:31-42 */
enum {
  NVBX_WEIGHT_CONSTANT = 0, NVBX_WEIGHT_CONSTANT_DROPOFF = 1, NVBX_WEIGHT_INVERSE_SQUARE = 2,
  NVBX_WEIGHT_INVERSE_SQUARE_DROPOFF = 3, NVBX_WEIGHT_INVERSE_SQUARE_TSDF_DISTANCE_PENALTY = 4,
  NVBX_WEIGHT_LINEAR_WITH_MAX = 5
};

/* The MapperParams fields that reach this path (mapper_initialization.cpp:231-466, values fuser.yaml:24-42). */
typedef struct {
  float voxel_size;                       /* node param voxel_size, node_params.hpp:84 */
  float max_integration_distance_m;       /* projective_integrator_max_integration_distance_m */
  float truncation_distance_vox;          /* projective_integrator_truncation_distance_vox */
  float max_weight;                       /* projective_integrator_max_weight */
  int32_t weighting_mode;                 /* projective_integrator_weighting_mode */
  int32_t raycast_subsampling_factor;     /* raycast_subsampling_factor */
  float esdf_min_weight;                  /* esdf_integrator_min_weight */
  float esdf_max_site_distance_vox;       /* esdf_integrator_max_site_distance_vox */
  float esdf_max_distance_m;              /* esdf_integrator_max_distance_m */
  float esdf_slice_height;                /* esdf_slice_height */
  float esdf_slice_min_height;            /* esdf_slice_min_height */
  float esdf_slice_max_height;            /* esdf_slice_max_height */
  float mesh_min_weight;                  /* mesh_integrator_min_weight */
  int32_t mesh_weld_vertices;             /* mesh_integrator_weld_vertices */
  int32_t sphere_tracing_subsampling;     /* colour integrator synthetic-depth subsampling (4) */
  int32_t sphere_tracing_max_steps;       /* 100 */
  float sphere_tracing_max_ray_length_m;  /* 15 */
  float sphere_tracing_surface_eps_vox;   /* 0.1 */
  float tsdf_decay_factor;                /* tsdf_decay_factor */
  float tsdf_decayed_weight_threshold;    /* tsdf_decayed_weight_threshold */
  int32_t esdf_site_rule;                 /* 0: inside && |d| <= max_site_distance (default); 1: |d| <= max_site_distance */
  int32_t depth_interp_nearest;           /* 0: bilinear with validity (default); 1: nearest */
  float lidar_max_integration_distance_m; /* lidar_projective_integrator_max_integration_distance_m (mapper_initialization.cpp:271-276) */
  float lidar_linear_interpolation_max_allowable_difference_vox;    /* [U] 2.0: bilinear taps must agree within this */
  float lidar_nearest_interpolation_max_allowable_dist_to_ray_vox;  /* [U] 0.5: nearest-beam fallback acceptance */
  int32_t workspace_bounds_type;          /* 0 unbounded, 1 height_bounds, 2 bounding_box (mapper_initialization.cpp:62-80,337-358) */
  float workspace_bounds_min_corner_m[3]; /* workspace_bounds_min_corner_{x,y}_m, workspace_bounds_min_height_m */
  float workspace_bounds_max_corner_m[3]; /* workspace_bounds_max_corner_{x,y}_m, workspace_bounds_max_height_m */
  int32_t do_depth_preprocessing;         /* do_depth_preprocessing (mapper_initialization.cpp:238-243; false in every shipped config) */
  int32_t depth_preprocessing_num_dilations; /* depth_preprocessing_num_dilations: invalid-depth regions grow by this many pixels */
  float invalid_depth_decay_factor;       /* projective_tsdf_integrator_invalid_depth_decay_factor (mapper_initialization.cpp:294-300;
                                             -1 = off, nvblox_base.yaml:80; 0.8 in nvblox_dynamics.yaml:11) */
  /* -- occupancy mappers: Mapper(voxel_size, memory_type, ProjectiveLayerType::kOccupancy) -- mapping_type static_occupancy
   *    (nvblox_base.yaml:9) and the dynamic / human mapper (specializations/nvblox_segmentation.yaml:9-22) */
  int32_t projective_layer_type;          /* 0 = TSDF (default), 1 = occupancy (log-odds), 2 = TSDF with a freespace layer */
  float free_region_occupancy_probability;      /* mapper_initialization.cpp:309-312; 0.45 nvblox_base.yaml:82 */
  float occupied_region_occupancy_probability;  /* :315-317; 0.55 */
  float unobserved_region_occupancy_probability;/* :320-322; 0.5 */
  float occupied_region_half_width_m;           /* :327; 0.1 */
  float free_region_decay_probability;          /* occupancy decay, :416; 0.55 */
  float occupied_region_decay_probability;      /* :421; 0.30 in the shipped configs */
  int32_t esdf_mode;                      /* node param esdf_mode, node_params.hpp:90: 0 = "2d" (default, the slice), 1 = "3d" (every voxel
                                             of every updated block; MultiMapper(voxel_size, mapping_type, EsdfMode::k3D, ...), nvblox_node.cpp:187-190) */
  /* -- freespace integrator of a projective_layer_type 2 mapper (MappingType::kDynamic's static mapper = TSDF with freespace,
   *    layer_publishing.cpp:747; parameters mapper_initialization.cpp:430-462, values nvblox_dynamics.yaml:12-18) */
  float max_tsdf_distance_for_occupancy_m;                    /* 0.15 */
  int32_t max_unobserved_to_keep_consecutive_occupancy_ms;    /* 200 */
  int32_t min_duration_since_occupied_for_freespace_ms;       /* 1000 (250 in nvblox_dynamics.yaml) */
  int32_t min_consecutive_occupancy_duration_for_reset_ms;    /* 2000 */
  int32_t check_neighborhood;                                 /* 1 */
  int32_t initialize_to_high_confidence_freespace;            /* 0 */
  /* -- [U] open choices (SURVEY.md 8a "open choices the oracle must expose as switches"): the arithmetic lives in the absent
   *    nvblox core, so every recollection that could be wrong is a switch implemented on BOTH sides (these kernels and
   *    oracle/nvblox_oracle.c) and parity-tested in both positions -- pinning to the real core is then a flag flip.
   *    0 is always the default behaviour documented in DESIGN.md section 3. */
  int32_t tsdf_weighting_variant;          /* formulas of the four non-trivial WeightingFunctionType modes: 0 = set A, 1 = set B */
  int32_t tsdf_skip_at_negative_truncation;/* voxel exactly at sdf == -truncation: 0 = integrated (skip iff sdf < -trunc), 1 = skipped (skip iff sdf <= -trunc) */
  int32_t tsdf_weight_clamp_before_blend;  /* 0 = blend with the unclamped sum, then w = min(w + w_m, max_weight); 1 = sum clamped first,
                                              distance blended as a running average over the clamped sum */
  float color_occlusion_threshold_vox;     /* colour occlusion test |synthetic depth - voxel depth| <= this (voxels); < 0 = the truncation distance */
  int32_t esdf_propagation;                /* 0 = exact Euclidean distance transform; 1 = synchronous 4-neighbour parent propagation to its fixed
                                              point, restricted to allocated ESDF blocks (the reference's sweep / propagate loop class) */
  int32_t mesh_ambiguity_rule;             /* marching-cubes ambiguous faces: 0 = inside corners cut off separately, 1 = inside corners joined,
                                              2 = complement-symmetric table (rule 0 for <= 4 inside corners, else rule 1: the classic table's behaviour) */
  int32_t mesh_normal_rule;                /* welded vertex normal: 0 = normal of the first triangle referencing it, 1 = area-weighted mean of the
                                              block's triangles referencing it */
  /* -- decay integrator switches (mapper_initialization.cpp:383-428; nvblox_base.yaml:103-107).  [U] semantics, DESIGN.md 3 */
  int32_t decay_deallocate_decayed_blocks; /* decay_integrator_deallocate_decayed_blocks (1): 0 = a fully decayed block stays allocated with its decayed voxels */
  int32_t tsdf_set_free_distance_on_decayed; /* tsdf_set_free_distance_on_decayed (0): 1 = an observed voxel whose weight falls below the threshold
                                              becomes FREE instead of fading to unknown: distance = tsdf_decayed_free_distance_vox voxels, weight = the threshold */
  float tsdf_decayed_free_distance_vox;    /* tsdf_decayed_free_distance_vox (4.0) */
  int32_t occupancy_decay_to_free;         /* occupancy_decay_to_free (0): 1 = occupied voxels decay past unknown into free and stay there; free voxels are not decayed */
  /* -- ground-plane-relative 2-D slice (mapper_initialization.cpp:136,257-260; [U] semantics).  With esdf_use_ground_plane the z
   *    band a column (x, y) of the slice looks at is [h + slice_height_above_plane_m, h + slice_height_above_plane_m + slice_height_thickness_m],
   *    h = the height of esdf_ground_plane {nx, ny, nz, d: n . p + d = 0, nz > 0} at the column's centre, instead of the fixed
   *    [esdf_slice_min_height, esdf_slice_max_height]; the output plane stays esdf_slice_height.  nvblox::MultiMapper sets plane and switch from its
   *    ground-plane estimator before every updateEsdf when multi_mapper.experimental_use_ground_plane_estimation is on. */
  float slice_height_above_plane_m;        /* slice_height_above_plane_m (0) */
  float slice_height_thickness_m;          /* slice_height_thickness_m (0) */
  int32_t esdf_use_ground_plane;           /* (0) */
  float esdf_ground_plane[4];
} nvbx_mapper_params;

/* nvblox::Lidar(num_azimuth_divisions, num_elevation_divisions, min_valid_range_m, vertical_fov_rad) or
 * (…, min_angle_below_zero_elevation_rad, max_angle_above_zero_elevation_rad) -- nvblox_node.cpp:1315-1323.
 * Elevations are signed here: min_elevation_rad < 0 below the horizon (vfov form: -vfov/2, +vfov/2). */
typedef struct {
  int32_t num_azimuth_divisions, num_elevation_divisions;
  float min_valid_range_m, min_elevation_rad, max_elevation_rad;
} nvbx_lidar;

/* Per-frame work counters (bench.py turns them into algorithmic bytes). */
typedef struct {
  int64_t blocks_allocated;     /* live hash entries */
  int64_t tsdf_blocks_in_view;  /* N_v of the last integrateDepth */
  int64_t color_blocks_updated; /* N_c of the last integrateColor */
  int64_t esdf_columns_marked;  /* N_u columns re-marked by the last updateEsdf */
  int64_t esdf_blocks_swept;    /* N_e ESDF blocks rewritten by the last updateEsdf */
  int64_t esdf_window_voxels;   /* voxels of the EDT working window (incl. halo) */
  int64_t mesh_blocks_updated;  /* N_m of the last updateColorMesh */
  int64_t mesh_vertices;        /* V written by the last updateColorMesh */
  int64_t mesh_triangles;       /* T (triangles) written by the last updateColorMesh */
  int64_t capacity_overflow;    /* != 0 if any pool / arena / window overflowed since creation */
  int64_t lidar_blocks_beam_centric; /* blocks in view of the last LiDAR scan that the beam-centric far-field launch updated (the dense launch took the rest) */
} nvbx_counters;

/* ---- lifetime --------------------------------------------------------------------------------------------------
 * nvblox::MultiMapper(voxel_size, MappingType, EsdfMode, MemoryType::kDevice, shared_ptr<CudaStream>)
 *   nvblox_ros/src/lib/nvblox_node.cpp:187-190, fuser_node.cpp:85-89.  `hip_stream` may be NULL (library-owned stream).
 * block_capacity = number of 8^3 blocks the HBM pools are sized for initially (3 x 4 KiB per block), 64 .. 2^24, or 0 = automatic
 * (about 4 % of the free HBM, 2^16 .. 2^20 blocks); the pools grow on demand, see nvbx_mapper_set_max_capacity.
 * Parameter values the kernels' loop bounds rely on are checked here and in nvbx_mapper_set_params (NVBX_E_INVALID, reason in
 * nvbx_last_error): voxel_size, both integration distances, truncation distance and max_weight > 0 and finite,
 * projective_layer_type in 0..2, esdf_mode in 0..1, occupancy probabilities strictly inside (0, 1). */
int nvbx_mapper_create(int device, void* hip_stream, const nvbx_mapper_params* params, int64_t block_capacity,
                       nvbx_mapper** out);
int nvbx_mapper_destroy(nvbx_mapper* m);
/* Pool growth.  The reference's layers allocate blocks on demand (Layer::allocateBlockAtIndex -- test_esdf_and_gradient_conversions.cpp:87);
 * here `block_capacity` is the INITIAL size of the HBM pools: before a frame is enqueued the integrate calls look at the number of free
 * slots the GPU last reported (pinned host memory, no synchronisation) and, below half, double every pool (contents kept, hash rebuilt
 * on the device; the call synchronises once) up to max_block_capacity (default 2^22, NVBX_MAX_BLOCKS in the environment; at most 2^24).
 * max_block_capacity <= the current capacity: fixed pools -- an allocation beyond them is dropped and reported through
 * nvbx_counters::capacity_overflow, never fatal.  Device pointers handed out by nvbx_get_device_view are valid until the next call
 * that can allocate, or that decays the map (the hash table of the surviving blocks is built in a second buffer: nvblox_hip_device.h).  nvbx_mapper_capacity = the current capacity. */
int nvbx_mapper_set_max_capacity(nvbx_mapper* m, int64_t max_block_capacity);
int64_t nvbx_mapper_capacity(nvbx_mapper* m);
/* the offline fuser's parameter values (nvblox_examples_bringup/config/nvblox/fuser.yaml:24-42; decay / workspace values from
 * nvblox_base.yaml:87-107) = the benchmark configuration; pure host function */
void nvbx_default_params(nvbx_mapper_params* out);
/* MultiMapper::setMapperParams  -- nvblox_node.cpp:203, fuser_node.cpp:94 */
int nvbx_mapper_set_params(nvbx_mapper* m, const nvbx_mapper_params* params);
int nvbx_mapper_get_params(const nvbx_mapper* m, nvbx_mapper_params* out);
/* CudaStream::synchronize -- conversions/esdf_slice_conversions.cu:107-108 */
int nvbx_synchronize(nvbx_mapper* m);
/* Enqueue everything the mapper holds back (the distance transform of the last nvbx_update_esdf, see there) on its stream
 * WITHOUT waiting: for callers that order their own work behind the mapper's with stream events instead of a host sync. */
int nvbx_flush(nvbx_mapper* m);
/* Colour deferral (cross-frame pipelining).  enable = 2, the DEFAULT of a new mapper since round 4: nvbx_integrate_color / _bgra8 of a single frame
 * is HELD BACK -- the image is copied into staging memory the mapper owns (one launch on the mapper's stream, ~0.9 MB at 640x480), the other arguments
 * are remembered -- and so is an nvbx_update_esdf that follows it.  The next single-frame nvbx_integrate_depth / _u16mm (for a held-back
 * nvbx_integrate_color_batch of n > 1 frames: the next nvbx_integrate_depth_batch of n > 1 frames) carries them out in pipelined order, two launches per
 * depth + colour + ESDF frame instead of four: {view marking of the new depth frame || sphere tracing, candidate-block discovery and ESDF site marking of
 * the held-back frame}, then {TSDF update of the new frame || colour integration and distance transform of the held-back frame} (three launches where the
 * mapper does not run the exact 2-D ESDF, or has integrated a LiDAR scan).  The overlapped parts are independent (DESIGN.md 2.8, table of invariants).
 * EVERY other entry point (queries, synchronize / flush, batches, LiDAR, mesh, decay, clearing, another integrate_color, ...) first carries the held-back
 * calls out exactly as they would have run at call time, so results -- map contents, ESDF, views, every query -- are bit-identical to the undeferred
 * sequence (tests/test_gpu_pipeline.py; the whole GPU suite runs with this default), and the caller may overwrite or free its colour image as soon as
 * nvbx_integrate_color has returned: nothing a host can observe through this header differs from enable = 0, except the time.  (What CAN tell the
 * difference: a kernel of the caller's own that reads the colour layer or the ESDF through include/nvblox_hip_device.h without calling nvbx_flush first.)
 * An nvbx_update_esdf that follows a depth frame WITHOUT a colour frame in between is held back the same way and carried out by the next camera depth
 * frame.  Argument errors are still reported by the call that made them.
 * enable = 1 (opt-in, zero-copy): no staging copy -- and a CONTRACT instead: the colour image passed to nvbx_integrate_color must stay valid and UNCHANGED
 * until the next call into this mapper has returned that either consumes or flushes the frame -- any call except nvbx_detect_dynamics,
 * nvbx_remove_small_components, nvbx_split_depth_by_mask, nvbx_dynamic_depth_split and nvbx_set_time_ms, which leave held-back work alone (the
 * dynamic-mapping frame starts with them).  nvblox_ros re-uses ONE colour buffer per node (nvblox_node.hpp:485-488) and fills it right before
 * integrateColor, after the depth frame -- compatible with the contract for the depth -> colour -> updateEsdf order of NvbloxNode::tick(); a node that
 * fills the colour buffer BEFORE calling integrateDepth must not use it.
 * enable = 0: the classic order, every call launches its own kernels (four launches per frame).  NVBX_COLOR_DEFERRAL=0|1|2 in the environment sets the
 * default of mappers created afterwards. */
int nvbx_mapper_set_color_deferral(nvbx_mapper* m, int32_t enable);
/* ---- Library-owned device frames: OWNERSHIP TRANSFER of input images (csrc/frames.hip).
 * Replaces the node-owned, re-used device image buffers of the reference (nvblox_node.hpp:484-488: color_image_, filled by the conversion on the
 * mapper's stream right before integrateColor, nvblox_node.cpp:1237-1263, conversions/image_conversions_thrust.cu:68-84) for a library that may hold a
 * colour frame back (nvbx_mapper_set_color_deferral above): an image that lives in a frame of nvbx_frame_acquire is RETAINED by the mapper instead of
 * copied (no k_stage_color launch, in either deferral form), and let go of when the launches that read it are enqueued.  A writer asks for "a frame
 * nobody else holds" before it writes: nvblox::Image<T> does so in its non-const dataPtr() / copyFromAsync / resize (include/nvblox/sensors/image.h) and
 * rotates to another frame while the mapper still holds the last one.
 *   nvbx_frame_acquire(device, bytes, writer_stream, &ptr): a device buffer of >= bytes, reference count 1 (the caller's).  Never hands out memory
 *     that launches of a mapper may still read: a frame a mapper has let go of is given to a writer only when those launches are known to have
 *     finished, or when `writer_stream` is the very stream they were enqueued on (stream order does the rest); otherwise another frame is taken, the
 *     pool grows (NVBX_FRAME_POOL_MAX frames per size, default 8) or -- pool full -- the call waits for the oldest.  writer_stream: the stream the
 *     caller will write the frame on, or NVBX_STREAM_UNKNOWN (host writes, any stream).
 *   nvbx_frame_retain / nvbx_frame_release: reference counting; at 0 the frame returns to the pool -- and is handed out again only after the work that
 *     may still use it: the release that brings the count to 0 records an event on every stream the library knows on the frame's device (the streams
 *     of live mappers and the legacy default stream), and nvbx_frame_acquire skips the frame until those events are reached (hipFree, which the
 *     reference's image buffers end in, waits for the device instead).  Only work on a non-blocking stream that no live mapper runs on is the
 *     caller's to order (or to name: nvbx_frame_release_on).  nvbx_frame_release_on(ptr, stream): the same for a caller that knows
 *     the ONE stream its work on the frame was enqueued on (one event; NVBX_STREAM_UNKNOWN = nvbx_frame_release).  A holder that lets go while others
 *     still hold the frame records nothing: its writes precede the others' reads by its own ordering, as with any shared buffer.
 *     nvbx_frame_refcount: > 1 = somebody else (a mapper) holds it too; -1 = not a frame.  nvbx_frame_device: the device the frame lives on (-1 = not a frame).
 *     nvbx_frame_writable: the question a writer asks before it overwrites a frame it holds (below).
 *   nvbx_frame_pool_trim(device | -1): hipFree every frame nobody holds; returns how many.  nvbx_frame_pool_stats: {held, free, bytes, created, waits, syncs}.
 *   nvbx_frame_upload: hipMemcpyAsync into a frame on `hip_stream` (NVBX_STREAM_UNKNOWN: a blocking hipMemcpy) -- for hosts without a HIP binding of
 *     their own (the ctypes mirror).
 *   nvbx_color_image_acquire(m, rows, cols, 3 | 4, &ptr) = nvbx_frame_acquire(device of m, rows * cols * bytes_per_pixel, stream of m, &ptr): for the
 *     converter that runs on the mapper's stream.  nvbx_integrate_color_owned(m, frame, 3 | 4, ...) = nvbx_integrate_color / _bgra8 + the caller's
 *     reference passes to the mapper (released for the caller; on NVBX_E_INVALID the caller keeps it). */
#define NVBX_STREAM_UNKNOWN ((void*)(intptr_t)-1)
int nvbx_frame_acquire(int device, size_t bytes, void* writer_stream, void** dev_ptr_out);
int nvbx_frame_retain(void* dev_ptr);
int nvbx_frame_release(void* dev_ptr);
int nvbx_frame_release_on(void* dev_ptr, void* last_stream);
int32_t nvbx_frame_device(const void* dev_ptr);
int32_t nvbx_frame_refcount(const void* dev_ptr);
/* 1 = the caller is the only holder AND no launch of a mapper (nor work recorded at an earlier release) can still be using the frame (finished, or enqueued on `writer_stream` itself): write in
 * place; 0 = take another frame (nvbx_frame_acquire) and let go of this one; -1 = not a live frame. */
int32_t nvbx_frame_writable(const void* dev_ptr, void* writer_stream);
int nvbx_frame_pool_trim(int device);
int nvbx_frame_pool_stats(int64_t out[6]);
int nvbx_frame_upload(void* dev_ptr, const void* src, size_t bytes, void* hip_stream);
int nvbx_color_image_acquire(nvbx_mapper* m, int32_t rows, int32_t cols, int32_t bytes_per_pixel, void** dev_ptr_out);
int nvbx_integrate_color_owned(nvbx_mapper* m, void* frame, int32_t bytes_per_pixel, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera);
/* The hipStream_t all of the mapper's work is enqueued on (the one handed to nvbx_mapper_create, or the library-owned one):
 * implicit conversion of nvblox::CudaStream to cudaStream_t -- conversions/esdf_slice_conversions.cu:107-108.  A caller that
 * reads / writes buffers it shares with the mapper on ANOTHER stream (e.g. an RCCL collective on the framework's stream) orders
 * the two with events on this handle. */
int nvbx_get_stream(nvbx_mapper* m, void** hip_stream_out);
/* Stream order between two mappers on DIFFERENT streams of one device (no reference counterpart: nvblox::MultiMapper hands both of its mappers one
 * CudaStream, nvblox_node.cpp:187-190; a host that creates its mappers on streams of their own -- stream = NULL at nvbx_mapper_create -- and passes
 * device images from one to the other, e.g. the split depth image of nvbx_dynamic_depth_split, orders them with this instead of a host
 * synchronisation): work enqueued on `waiter`'s stream after this call starts only when
 * everything ENQUEUED on `producer`'s stream before it has finished (held-back calls of `producer` are not enqueued yet and are not waited for).
 * Asynchronous; a no-op when both run on one stream. */
int nvbx_mapper_wait_for(nvbx_mapper* waiter, nvbx_mapper* producer);
const char* nvbx_last_error(void);
/* Arithmetic self-test (no reference counterpart; test infrastructure of the boundary): evaluates the library's own division and
 * square-root sequences (csrc/nvbx_arith.h: the compiler's IEEE sequences without their range-scaling steps) on device arrays --
 * quot[i] = a[i] / b[i], root[i] = sqrt(|a[i]|) -- on the current device's null stream and waits.  tests/test_gpu_arith.py compares
 * the results with IEEE float division / square root bit for bit: the CPU oracle uses the plain operators. */
int nvbx_selftest_arith(const float* a_dev, const float* b_dev, float* quot_dev, float* root_dev, int64_t n);
/* Mapper::clear / fresh map (load_map path re-creates the mapper: nvblox_node.cpp:1698-1703) */
int nvbx_mapper_clear(nvbx_mapper* m);

/* ---- integration (asynchronous on the mapper stream) --------------------------------------------------------------
 * Argument checks (NVBX_E_INVALID, nothing launched): the camera's width / height must equal the image's cols / rows and its
 * focal lengths be > 0; image sides are 1 .. 32768 (pixel indices are 31-bit on the device); T_L_C must be finite and keep everything within the integration distance inside the addressable block
 * range (nvbx_index3d).  Every entry point makes the mapper's device the calling thread's current device and leaves it so.
 * MultiMapper::integrateDepth(const DepthImage&, const Transform& T_L_C, const Camera&, Time) -- nvblox_node.cpp:1062 */
int nvbx_integrate_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                         const nvbx_camera* camera);
/* Same, depth given as uint16 millimetres: fuses conversions::depthImageFromNitrosViewAsync's DivideBy1000
 * (conversions/image_conversions_thrust.cu:39-45,140-142) into the integrator's depth read. */
int nvbx_integrate_depth_u16mm(nvbx_mapper* m, const uint16_t* depth_mm_dev, int32_t rows, int32_t cols,
                               const float T_L_C[16], const nvbx_camera* camera);
/* MultiMapper::integrateDepth(const Pointcloud&, const Transform&, const Lidar&, ...) -- nvblox_node.cpp:1382-1384, after
 * the point cloud has been rendered to a range image (nvbx_depth_image_from_pointcloud; the reference exposes that image
 * as getLastDepthFrameFromPointcloud(), nvblox_node.cpp:1397).  range_dev: rows = elevation divisions, cols = azimuth
 * divisions, metres along the beam, <= 0 invalid.  Uses lidar_max_integration_distance_m. */
int nvbx_integrate_lidar_depth(nvbx_mapper* m, const float* range_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                               const nvbx_lidar* lidar);
/* depthImageFromPointcloudKernel (conversions/pointcloud_conversions.cu:118-150): points (x,y,z f32, sensor frame) ->
 * range image; NaN points and points outside the model are skipped; several points in one pixel: one of them wins. */
int nvbx_depth_image_from_pointcloud(nvbx_mapper* m, const float* points_xyz_dev, int64_t n_points, const nvbx_lidar* lidar,
                                     float* range_dev);
/* [U] LiDAR motion compensation (use_lidar_motion_compensation, nvblox_node.cpp:1339-1384): every point carries its time within the
 * scan (rel_time_ms, 0 = scan start = the time T_L_S_start refers to); the sensor pose at that time is interpolated between
 * T_L_S_start and T_L_S_end (translation linearly, rotation by normalised quaternion interpolation) and the point is re-expressed
 * in the sensor frame at scan start.  Out may alias in.  Asynchronous.  Feed the result to nvbx_depth_image_from_pointcloud. */
int nvbx_motion_compensate_pointcloud(nvbx_mapper* m, const float* points_in_dev, const float* rel_time_ms_dev, int64_t n_points,
                                      const float T_L_S_start[16], const float T_L_S_end[16], float scan_duration_ms, float* points_out_dev);
/* MultiMapper::integrateColor(const ColorImage&, const Transform&, const Camera&) -- nvblox_node.cpp:1264.
 * rgb_dev: rows*cols*3 bytes, nvblox::Color order r,g,b (image_conversions.cpp:100-101). */
int nvbx_integrate_color(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                         const nvbx_camera* camera);
/* Same, colour given as bgra8 (4 bytes per pixel): fuses conversions::colorImageFromNitrosViewAsync's ToRgba<Bgra>
 * channel reorder (conversions/image_conversions_thrust.cu:60-65, image_conversions.cpp:170-176) into the integrator's
 * colour fetch (one aligned 4-byte load per tap). */
int nvbx_integrate_color_bgra8(nvbx_mapper* m, const uint8_t* bgra_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                               const nvbx_camera* camera);
/* ---- camera batches: the reference's multi-camera mode (up to four cameras feed ONE mapper through one queue, an integrateDepth /
 * integrateColor call per camera frame: nvblox_ros/include/nvblox_ros/nvblox_node.hpp:298-332, nvblox_node.cpp:247-292) as ONE launch
 * set.  n (1 .. NVBX_MAX_BATCH) frames of the same image size; depth_dev / rgb_dev = n device image pointers (host array), T_L_C =
 * n x 16 floats, cameras = n structs.  DEFINED as equal to the n separate calls in order 0 .. n-1 -- map contents bit for bit, the
 * "last view" queries (nvbx_last_depth_view / _color_view, decayTsdfExcludeLastView, nvbx_get_synthetic_depth) report camera n-1's --
 * but with one view-marking + one TSDF-update launch (depth) and one sphere-tracing + one colour launch for all n cameras: at
 * 640x480 a frame is launch / latency bound, so a batch of 8 costs far less than 8 frames.  Mappers with a freespace layer and depth
 * dilation fall back to the separate calls. */
#define NVBX_MAX_BATCH 8
int nvbx_integrate_depth_batch(nvbx_mapper* m, int32_t n, const float* const* depth_dev, int32_t rows, int32_t cols, const float* T_L_C,
                               const nvbx_camera* cameras);
int nvbx_integrate_color_batch(nvbx_mapper* m, int32_t n, const uint8_t* const* rgb_dev, int32_t rows, int32_t cols, const float* T_L_C,
                               const nvbx_camera* cameras);
/* ---- a PAIR of mappers, one depth frame each: MultiMapper::integrateDepth of the dynamic and the human mapping types (nvblox_node.cpp:1057-1062:
 * the depth image is split by a mask; the unmasked part goes to the background mapper, the masked part to the foreground occupancy mapper, one
 * integrateDepth each).  DEFINED as equal to nvbx_integrate_depth(ma, depth_a, ...) followed by nvbx_integrate_depth(mb, depth_b, ...) -- both maps, the
 * held-back calls each of them carries, "last view" queries: bit for bit -- but the two view-marking launches share one grid and so do the two
 * TSDF-update launches (k_mark_view_pair, k_integrate_tsdf_color_pair): two launches instead of four; the maps share nothing, so nothing else changes.
 * Needs both mappers on one device and one stream (what nvblox::MultiMapper hands out); whatever the pair cannot express -- different streams, depth
 * dilation, a held-back colour BATCH, the ESDF side stream -- falls back to the two calls.  NVBX_DEPTH_PAIR=0 in the environment: always the two calls. */
int nvbx_integrate_depth_pair(nvbx_mapper* ma, const float* depth_a_dev, nvbx_mapper* mb, const float* depth_b_dev, int32_t rows, int32_t cols,
                              const float T_L_C[16], const nvbx_camera* camera);
/* MultiMapper::updateEsdf() (EsdfMode::k2D) -- nvblox_node.cpp:781.
 * Scheduling note: the site-marking half runs at once (or already ran inside the preceding nvbx_integrate_color launch);
 * the distance-transform half may be HELD BACK until the next entry point of this mapper: nvbx_integrate_depth[_u16mm]
 * runs it inside its first launch, beside the view marking it does not interact with; every other entry point -- all
 * queries, nvbx_synchronize, colour / LiDAR integration, mesh, maintenance -- enqueues it first.  Results observed through
 * this API are therefore always those of the completed update; a caller that only synchronises the raw stream and calls
 * nothing else leaves the transform un-enqueued until its next call.  NVBX_DEFER_EDT=0 in the environment disables this. */
int nvbx_update_esdf(nvbx_mapper* m);
/* Mapper::updateColorMesh(UpdateFullLayer) -- layer_publishing.cpp:686-689, nvblox_node.cpp:1611 */
int nvbx_update_color_mesh(nvbx_mapper* m, int32_t update_full_layer);
/* Mapper::decayTsdfExcludeLastView<Camera>() / decayTsdf -- nvblox_node.cpp:931-936 */
int nvbx_decay_tsdf(nvbx_mapper* m, int32_t exclude_last_view);
/* Mapper::decayOccupancyAllVoxels() -- nvblox_node.cpp:925-929 (occupancy mappers): log-odds move towards 0 (unknown) by the
 * log-odds of free_region_decay_probability / occupied_region_decay_probability and stop there; all-unknown blocks are deallocated */
int nvbx_decay_occupancy(nvbx_mapper* m);
/* Mapper::clearOutsideRadius(center, radius) -- nvblox_node.cpp:1566-1583 */
int nvbx_clear_outside_radius(nvbx_mapper* m, const float center[3], float radius);

/* Mapper::getClearedBlocks(layer_types) -- layer_publishing.cpp:716,804: Index3D (sorted, unique) of the projective-layer (TSDF /
 * occupancy) blocks that decayTsdf / decayOccupancyAllVoxels / clearOutsideRadius deallocated since the last call; the list is
 * emptied.  Returns their number n; if n > capacity nothing is written and the list is kept (call again with room for n);
 * synchronises. */
int64_t nvbx_take_cleared_blocks(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity);

/* Mapper::clearTsdfInsideShapes(std::vector<BoundingShape>) -- nvblox_node.cpp:1834 (the EsdfAndGradients service's
 * clearing request, conversions/esdf_and_gradients_conversions.cu:127-180).  A shape is a sphere (kind 0: centre, radius)
 * or an axis-aligned box (kind 1: min corner, max corner); TSDF voxels whose centre lies inside any shape are reset to
 * unobserved (distance 0, weight 0) and their blocks become ESDF / mesh dirty. */
typedef struct { int32_t kind; float a[3]; float b[3]; } nvbx_bounding_shape;   /* sphere: a = centre, b[0] = radius; box: a = min, b = max */
int nvbx_clear_tsdf_inside_shapes(nvbx_mapper* m, const nvbx_bounding_shape* shapes_host, int32_t n_shapes);

/* ---- ESDF slice (EsdfSlicer) ------------------------------------------------------------------------------------
 * EsdfSlicer::sliceLayerToDistanceImage(esdf_layer, slice_height, unknown_value, &aabb, &Image<float>) --
 * nvblox_node.cpp:836-844.  Two-step like Image<float> allocation: query the size, then fill a device image
 * (row = y, col = x, row-major; origin aabb[0..2]; conversions/esdf_slice_conversions.cu:60-64). */
int nvbx_esdf_slice_size(nvbx_mapper* m, int32_t* rows, int32_t* cols, float aabb_min_max[6]);
int nvbx_esdf_slice_to_image(nvbx_mapper* m, float unknown_value, float* image_dev, int64_t capacity_elems,
                             int32_t* rows, int32_t* cols, float aabb_min_max[6]);
/* EsdfSliceConverter::distanceMapSliceMsgFromSliceImage's D2H (esdf_slice_conversions.cu:81-109) in one call -- and ONE wait for the device: the
 * slicing launch reads the layer's AABB itself and writes size + image into pinned host memory (no size query first; round 5).  NVBX_E_CAPACITY
 * (image_host null or capacity_elems too small): *rows / *cols / aabb report what is needed, nothing is copied. */
int nvbx_esdf_slice_to_host(nvbx_mapper* m, float unknown_value, float* image_host, int64_t capacity_elems,
                            int32_t* rows, int32_t* cols, float aabb_min_max[6]);
/* EsdfSlicer::sliceLayersToCombinedDistanceImage(layer_1, layer_2, height_1, height_2, unknown, &aabb, &image) --
 * nvblox_node.cpp:836-840 (static + dynamic mapper in one costmap): the image covers the union of the two layers' AABBs,
 * a pixel is the smaller of the two signed distances where both are observed, the observed one where only one is, else
 * unknown_value.  Both mappers must live on the same device; the work is enqueued on m1's stream behind everything
 * enqueued on m2's (event).  aabb z range = m1's slice block. */
int nvbx_esdf_slice_combined_size(nvbx_mapper* m1, nvbx_mapper* m2, int32_t* rows, int32_t* cols, float aabb_min_max[6]);
int nvbx_esdf_slice_combined_to_image(nvbx_mapper* m1, nvbx_mapper* m2, float unknown_value, float* image_dev, int64_t capacity_elems,
                                      int32_t* rows, int32_t* cols, float aabb_min_max[6]);
/* EsdfSlicer::occupancyGridFromSliceImage(img, int8_t*, unknown) -- nvblox_node.cpp:917-919:
 * 100 where distance <= 0 (inside), 0 where observed free, -1 where unknown. */
int nvbx_occupancy_grid_from_slice(nvbx_mapper* m, const float* image_dev, int32_t rows, int32_t cols,
                                   float unknown_value, int8_t* grid_dev);
/* conversions::EsdfSliceConverter::pointcloudFromSliceImage kernel (esdf_slice_conversions.cu:33-73,138-159):
 * compacts observed pixels to {x,y,z,intensity} float4; returns the count through *n_points (synchronises). */
int nvbx_pointcloud_from_slice(nvbx_mapper* m, const float* image_dev, int32_t rows, int32_t cols,
                               const float aabb_min_max[6], float slice_height, float unknown_value,
                               float* points_xyzi_dev, int32_t* n_points);
/* voxelLayerToDenseVoxelGridInAABBAsync<SignedDistanceFunctor> (esdf_and_gradients_conversions.cu:88-125):
 * out[x*(Ny*Nz) + y*Nz + z] = signed metres or default_value; AABB given in global voxel indices. */
int nvbx_esdf_dense_grid(nvbx_mapper* m, const int32_t min_vox[3], const int32_t size_vox[3], float default_value,
                         float* grid_dev);

/* ---- map file (Mapper::saveLayerCake(path) -> bool, loadMap(path) -> bool: nvblox_node.cpp:1663-1668,1698-1703) -----------------
 * A path ending in .nvblx is written as an SQLITE database, like the reference's layer cake: table layers(layer_type, voxel_size,
 * block_size, voxel_bytes, num_blocks) + one table <layer_type>_blocks(index_x, index_y, index_z, data BLOB) per layer (tsdf_layer,
 * color_layer, esdf_layer, occupancy_layer; data = 512 reference voxel structs).  [U] The schema is a guess at upstream's (the
 * serializer lives in the absent core): files open with any sqlite3 tool, interoperability with upstream's reader is unverified.
 * libsqlite3 is loaded at run time; any other path (or no libsqlite3) uses a compact little-endian container of our own.  load
 * recognises either by its magic and replaces the map: it is cleared first, the file's voxel size must equal the mapper's, loaded TSDF
 * blocks are ESDF- and mesh-dirty.  Errors: NVBX_E_IO, NVBX_E_INVALID (voxel size), NVBX_E_CAPACITY; a file that fails validation
 * leaves the current map untouched. */
int nvbx_save_map(nvbx_mapper* m, const char* path);
int nvbx_load_map(nvbx_mapper* m, const char* path);

/* ---- sensor-side conversions next to the path -----------------------------------------------------------------------
 * DepthImageBackProjector::backProjectOnGPU(depth, camera, &pointcloud_C, max_back_projection_distance) --
 * nvblox_node.cpp:1128-1130, fuser_node.cpp:294-296: every pixel with 0 < depth <= max_distance_m (max <= 0: no limit)
 * becomes the point ((u + 0.5 - cu) / fu * d, (v + 0.5 - cv) / fv * d, d) of the camera frame; the points are compacted
 * (order unspecified) into points_xyz_dev[capacity_points][3]; *n_points = their number (synchronises). */
int nvbx_backproject_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const nvbx_camera* camera,
                           float max_distance_m, float* points_xyz_dev, int64_t capacity_points, int64_t* n_points);
/* transformPointcloudOnGPU(T_L_C, pointcloud_C, &pointcloud_L) -- nvblox_node.cpp:1131, fuser_node.cpp:297 (in place if
 * out == in; asynchronous on the mapper's stream) */
int nvbx_transform_pointcloud(nvbx_mapper* m, const float T_L_C[16], const float* points_in_dev, int64_t n_points, float* points_out_dev);

/* ---- mask splitting (human / people-segmentation mapping) ------------------------------------------------------------
 * [U] ImageMasker::splitImageOnGPU as used by MultiMapper::integrateDepth(depth, mask, T_L_CD, T_CM_CD, depth_cam, mask_cam) --
 * nvblox_node.cpp:1018-1060: every valid depth pixel is lifted to 3-D, moved into the mask camera (T_CM_CD = T_L_CM^-1 T_L_CD)
 * and projected; it is MASKED if it lands on a non-zero mask pixel and is not occluded there (its depth in the mask camera is
 * within occlusion_threshold_m of the nearest depth pixel that landed on the same mask pixel).  depth_unmasked gets the pixel
 * if it is not masked, depth_masked if it is; the other image gets NVBX_MASKED_DEPTH_INVALID (-1, an invalid depth for the
 * integrators).  overlay_rgb (may be NULL): grey depth with masked pixels tinted red (debug image).  Asynchronous. */
#define NVBX_MASKED_DEPTH_INVALID (-1.0f)
int nvbx_split_depth_by_mask(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const uint8_t* mask_dev,
                             int32_t mask_rows, int32_t mask_cols, const float T_CM_CD[16], const nvbx_camera* depth_camera,
                             const nvbx_camera* mask_camera, float occlusion_threshold_m, float* depth_unmasked_dev,
                             float* depth_masked_dev, uint8_t* overlay_rgb_dev);
/* MultiMapper::integrateColor(color, mask, T_L_C, camera) -- nvblox_node.cpp:1261-1262 (mask and colour share the camera):
 * [U] masked pixels are black in rgb_unmasked and the only non-black ones in rgb_masked (either output may be NULL). */
int nvbx_split_color_by_mask(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const uint8_t* mask_dev,
                             uint8_t* rgb_unmasked_dev, uint8_t* rgb_masked_dev);

/* ---- dynamic mapping (MappingType::kDynamic, nvblox_dynamics.yaml) -----------------------------------------------------------
 * update_time_ms of MultiMapper::integrateDepth(depth, T_L_C, camera, update_time_ms) (nvblox_node.cpp:1062): host state, read by the
 * next nvbx_integrate_depth of a projective_layer_type 2 mapper, which then also updates the freespace layer of the blocks in view
 * ([U] FreespaceIntegrator, semantics in DESIGN.md 3). */
int nvbx_set_time_ms(nvbx_mapper* m, int64_t update_time_ms);
/* [U] DynamicsDetection::computeDynamics: mask_dev[rows][cols] (u8) = 1 where a valid depth pixel (<= max_distance_m if > 0) lies in a
 * high-confidence-freespace voxel, else 0.  Asynchronous. */
int nvbx_detect_dynamics(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                         float max_distance_m, uint8_t* mask_dev);
/* [U] removeSmallConnectedComponents (multi_mapper connected_mask_component_size_threshold, mapper_initialization.cpp:130): erases the
 * 8-connected components of non-zero pixels smaller than min_size.  In place; asynchronous on the mapper's stream (lock-free union-find). */
int nvbx_remove_small_components(nvbx_mapper* m, uint8_t* mask_dev, int32_t rows, int32_t cols, int32_t min_size);
/* The front end of MultiMapper::integrateDepth(depth, T_L_C, camera) in MappingType::kDynamic (nvblox_node.cpp:1062; its outputs are read at
 * :1098,1108) as ONE call: nvbx_detect_dynamics -> nvbx_remove_small_components(min_component_size; <= 0: not removed) ->
 * nvbx_split_depth_by_mask with the depth camera as the mask camera (T_CM_CD = identity).  Same results as the three calls, bit for bit
 * (mask_dev = the cleaned mask; overlay_rgb_dev may be NULL), in three launches and no memset instead of six and one.  Like
 * nvbx_detect_dynamics it leaves held-back work (colour deferral) alone.  Asynchronous. */
int nvbx_dynamic_depth_split(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                             float max_distance_m, int32_t min_component_size, float occlusion_threshold_m,
                             uint8_t* mask_dev, float* depth_unmasked_dev, float* depth_masked_dev, uint8_t* overlay_rgb_dev);

/* ---- device-side view for the caller's own kernels (GPULayerView / gpu_indexing.cuh: esdf_slice_conversions.cu:18,
 * esdf_and_gradients_conversions.cu:19-23).  Accessors: include/nvblox_hip_device.h. */
typedef struct {
  const void* table; uint32_t table_mask, table_shift;    /* open-addressing hash: 16-byte entries {key64, slot32, stamp32} */
  const uint32_t* slot_flags;                              /* per slot: NVBX_LAYER_* bits in the low byte */
  const int32_t* slot_index;                               /* per slot: block index x, y, z */
  const void* tsdf; const void* color; const void* esdf;   /* voxel pools, 512 x 8 bytes per slot */
  float voxel_size; uint32_t block_capacity;
} nvbx_device_view;
int nvbx_get_device_view(nvbx_mapper* m, nvbx_device_view* out);

/* ---- layer access (Layer<VoxelBlock> accessors; synchronise) ----------------------------------------------------
 * layer.numAllocatedBlocks(), getAllBlockIndices(), getBlockAtIndex(), allocateBlockAtIndex(),
 * callFunctionOnAllVoxels -- test_esdf_and_gradient_conversions.cpp:85-92,114,118 */
int64_t nvbx_num_blocks(nvbx_mapper* m, uint32_t layer);
int64_t nvbx_block_indices(nvbx_mapper* m, uint32_t layer, nvbx_index3d* out, int64_t capacity);
int nvbx_get_block(nvbx_mapper* m, uint32_t layer, nvbx_index3d idx, void* voxels_out /* 512 reference structs */);
int nvbx_set_block(nvbx_mapper* m, uint32_t layer, nvbx_index3d idx, const void* voxels_in);
/* batched allocateBlockAtIndex + whole-block write: voxels_in[n][512] reference structs (TSDF blocks become ESDF- and mesh-dirty) */
int nvbx_set_blocks(nvbx_mapper* m, uint32_t layer, const nvbx_index3d* idx, int64_t n, const void* voxels_in);
/* batched getBlockAtIndex: n blocks into voxels_out[n][512]; found_out[i] = 1 if block i exists (may be NULL) */
int nvbx_get_blocks(nvbx_mapper* m, uint32_t layer, const nvbx_index3d* idx, int64_t n, void* voxels_out, int32_t* found_out);
/* blocks touched by the last integrateDepth / integrateColor (what Mapper records as "blocks to update") */
int64_t nvbx_last_depth_view(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity);
int64_t nvbx_last_color_view(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity);
int nvbx_get_synthetic_depth(nvbx_mapper* m, float* out_host, int64_t capacity, int32_t* rows, int32_t* cols);
int nvbx_get_counters(nvbx_mapper* m, nvbx_counters* out);

/* ---- mesh output (SerializedColorMeshLayer accessors, conversions/mesh_conversions.cpp:62-104) -------------------
 * Flat arrays + per-block offsets for the blocks meshed by the last nvbx_update_color_mesh: vertices (x,y,z f32),
 * normals (f32 x3), colours (rgba u8), triangle indices (int32, local to the block). */
int nvbx_mesh_sizes(nvbx_mapper* m, int64_t* n_blocks, int64_t* n_vertices, int64_t* n_triangles);
int nvbx_mesh_copy(nvbx_mapper* m, nvbx_index3d* block_indices, int32_t* vertex_offsets /* n_blocks+1 */,
                   int32_t* triangle_offsets /* n_blocks+1 */, float* vertices, float* normals, uint8_t* colors,
                   int32_t* triangles);

/* The exchange message without an export launch: once a buffer is registered (host state only), every nvbx_integrate_depth /
 * _lidar_depth also writes the Index3D of the blocks it updates into it -- int32 [1 + capacity][3], row 0 = {count, 0, 0} -- from
 * inside its TSDF-update launch.  (The blocks of THIS depth frame; nvbx_esdf_dirty_list also covers decay / clearing / set_block.)
 * NULL unregisters.  The buffer must stay valid while registered. */
int nvbx_set_view_export(nvbx_mapper* m, int32_t* packed_dev, int64_t capacity);
/* ---- multi-GPU (one camera per GPU, all-gather of updated block indices before the ESDF sweep) -----
 * nvbx_esdf_dirty_list writes the Index3D of the TSDF blocks dirtied since the last updateEsdf into caller-owned
 * device buffers (indices int32[capacity][3], count int32[1], clamped to capacity) -- asynchronous, no host copy.
 * The caller all-gathers both over RCCL and feeds every peer's list back with nvbx_mark_esdf_dirty (count is read
 * on the device; max_count bounds it). */
int nvbx_esdf_dirty_list(nvbx_mapper* m, int32_t* indices_dev_out, int32_t* count_dev_out, int64_t capacity);
int nvbx_mark_esdf_dirty(nvbx_mapper* m, const int32_t* indices_dev, const int32_t* count_dev, int64_t max_count);
/* The same for the whole all-gathered buffer in ONE launch: gathered_dev is int32 [world][1 + max_count][3], row 0 of
 * each rank = (count, -, -), rows 1.. = block indices (the packed layout isaac_ros_nvblox_amd/dist.py all-gathers);
 * rank `self_rank`'s own list is skipped. */
int nvbx_mark_esdf_dirty_gathered(nvbx_mapper* m, const int32_t* gathered_dev, int32_t world, int32_t self_rank, int64_t max_count);
/* The same union step, held back -- no launch of its own.  WHEN it runs, and how long `gathered_dev` must stay valid and unchanged:
 *   classic launch order: extra workgroups of the next nvbx_integrate_color launch perform it beside the marking of the mapper's own
 *     dirty blocks; the peers' blocks reach the ESDF with the NEXT nvbx_update_esdf (one update after the frame that produced them).
 *   pipelined order (colour deferral, the default): it rides in the fused TSDF-update launch of the next nvbx_integrate_depth, where it only
 *     sets the blocks ESDF-dirty for the marking pass AFTER that -- the peers' blocks reach the ESDF TWO updates after the frame that produced
 *     them, and the buffer is read one nvbx_integrate_depth later than in classic order.
 *   any other entry point that comes first performs it at once (its own launch).
 * A caller that refills gathered buffers in rotation therefore needs THREE sets (filling / in flight in the collective / being read here):
 * nvblox::BlockIndexExchange (include/nvblox/mapper/block_index_exchange.h) and dist.PipelinedDirtyBlockExchange do exactly that; two sets race
 * with the rider.  Under option (A) of SURVEY.md 8e (replicas + index union) the extra update of delay changes no voxel: a peer's block only
 * re-marks a column from the local, unchanged TSDF. */
int nvbx_mark_esdf_dirty_gathered_deferred(nvbx_mapper* m, const int32_t* gathered_dev, int32_t world, int32_t self_rank, int64_t max_count);

/* ---- multi-GPU, one fused map (SURVEY.md 8e option B, made exact): measurement exchange ---------------------------------------------
 * One camera per GPU.  nvbx_measure_depth runs the view calculation and the per-voxel projection / depth sampling of THIS rank's camera
 * -- the expensive, sharded part of integrateDepth -- and writes, per block in view, a record {Index3D, 512 x {measured depth, voxel
 * depth}} (voxel depth < 0: voxel not touched; measured depth < 0: it projects onto invalid depth) plus the record count, into
 * caller-owned device buffers (asynchronous, no host copy).  The caller all-gathers the buffers (RCCL over xGMI; isaac_ros_nvblox_amd/
 * dist.py MeasurementFusion) and hands the gathered array [world][stride_blocks] + counts [world] to nvbx_apply_measurements on every
 * rank, which applies every camera's measurements to the local map in RANK ORDER with the integrator's own per-voxel update: every rank
 * then holds the SAME map, bit-identical to ONE mapper integrating the cameras in rank order (nvbx_integrate_depth_batch) -- which
 * exchanging already-fused {distance, weight} blocks cannot give (the clamps do not commute with a weighted mean).  owner_mod > 1: only
 * blocks with Index3DHash(block) mod owner_mod == owner_rank are applied (the rank's shard of that map; the other blocks stay empty).
 * world <= NVBX_MAX_BATCH.  Two launches per apply, whatever the world size. */
typedef struct { int32_t x, y, z, rank; float ds_vd[512][2]; } nvbx_measurement_block;     /* 4112 bytes */
int nvbx_measure_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                       nvbx_measurement_block* out_dev, int32_t* count_dev, int64_t capacity_blocks);
int nvbx_apply_measurements(nvbx_mapper* m, const nvbx_measurement_block* gathered_dev, const int32_t* counts_dev, int32_t world, int64_t stride_blocks,
                            int32_t owner_mod, int32_t owner_rank);

/* ---- ground plane (MultiMapper::ground_plane_estimator(), nvblox_node.cpp:1456,1474; parameters mapper_initialization.cpp:133-153) ------
 * [U] GroundPlaneEstimator restated.  nvbx_tsdf_zero_crossings: the upward zero crossings of the TSDF -- vertically adjacent observed voxels
 * with d(z) <= 0 < d(z + 1), interpolated linearly along z -- whose height lies in [min_z_m, max_z_m] (ground_points_candidates_min/max_z_m),
 * as xyz triples in HOST memory sorted by (x, y, z); returns the count (> capacity: nothing written, come back with room); synchronises.
 * nvbx_fit_plane_ransac (host only, no mapper): `iterations` (num_ransac_iterations) sampled triples, inliers within distance_threshold_m
 * (ransac_distance_threshold_m), normal turned upwards; plane_out = {nx, ny, nz, d} with n . p + d = 0; returns the inlier count of the winner
 * (0 = no plane).  The sampling sequence is fixed by `seed`, so the estimate is a pure function of its inputs. */
int64_t nvbx_tsdf_zero_crossings(nvbx_mapper* m, float min_z_m, float max_z_m, float* points_xyz_host, int64_t capacity);
int64_t nvbx_fit_plane_ransac(const float* points_xyz_host, int64_t n, float distance_threshold_m, int32_t iterations, uint32_t seed, float plane_out[4]);

/* ---- instrumentation (timing::Timer analogue for the per-kernel roofline line of bench.py) -----------------------
 * While enabled every kernel launch is bracketed by a hipEvent pair on the mapper stream. nvbx_get_profile returns a
 * JSON object {"kernel": {"count": n, "total_ms": t}}; the entry "_empty_event_pair" is the span of event pairs with nothing between
 * them = what the instrumentation itself adds to the span of one launch (subtract it once per launch). */
int nvbx_set_profiling(nvbx_mapper* m, int32_t enable);
int nvbx_get_profile(nvbx_mapper* m, char* json_out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* NVBLOX_HIP_H_ */
