"""Loader of libnvblox_hip.so (the C-ABI in include/nvblox_hip.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVBX_LIB") or os.path.join(_HERE, "libnvblox_hip.so")      # (NVBX_LIB: a tuning variant of the same library, tools/)


class Params(C.Structure):
    """nvbx_mapper_params (include/nvblox_hip.h); field names = the reference's ROS parameter names
    (nvblox_ros/src/lib/mapper_initialization.cpp:231-466) without the integrator prefixes."""
    _fields_ = [
        ("voxel_size", C.c_float),
        ("max_integration_distance_m", C.c_float),
        ("truncation_distance_vox", C.c_float),
        ("max_weight", C.c_float),
        ("weighting_mode", C.c_int32),
        ("raycast_subsampling_factor", C.c_int32),
        ("esdf_min_weight", C.c_float),
        ("esdf_max_site_distance_vox", C.c_float),
        ("esdf_max_distance_m", C.c_float),
        ("esdf_slice_height", C.c_float),
        ("esdf_slice_min_height", C.c_float),
        ("esdf_slice_max_height", C.c_float),
        ("mesh_min_weight", C.c_float),
        ("mesh_weld_vertices", C.c_int32),
        ("sphere_tracing_subsampling", C.c_int32),
        ("sphere_tracing_max_steps", C.c_int32),
        ("sphere_tracing_max_ray_length_m", C.c_float),
        ("sphere_tracing_surface_eps_vox", C.c_float),
        ("tsdf_decay_factor", C.c_float),
        ("tsdf_decayed_weight_threshold", C.c_float),
        ("esdf_site_rule", C.c_int32),
        ("depth_interp_nearest", C.c_int32),
        ("lidar_max_integration_distance_m", C.c_float),
        ("lidar_linear_interpolation_max_allowable_difference_vox", C.c_float),
        ("lidar_nearest_interpolation_max_allowable_dist_to_ray_vox", C.c_float),
        ("workspace_bounds_type", C.c_int32),
        ("workspace_bounds_min_corner_m", C.c_float * 3),
        ("workspace_bounds_max_corner_m", C.c_float * 3),
        ("do_depth_preprocessing", C.c_int32),
        ("depth_preprocessing_num_dilations", C.c_int32),
        ("invalid_depth_decay_factor", C.c_float),
        ("projective_layer_type", C.c_int32),
        ("free_region_occupancy_probability", C.c_float),
        ("occupied_region_occupancy_probability", C.c_float),
        ("unobserved_region_occupancy_probability", C.c_float),
        ("occupied_region_half_width_m", C.c_float),
        ("free_region_decay_probability", C.c_float),
        ("occupied_region_decay_probability", C.c_float),
        ("esdf_mode", C.c_int32),
        ("max_tsdf_distance_for_occupancy_m", C.c_float),
        ("max_unobserved_to_keep_consecutive_occupancy_ms", C.c_int32),
        ("min_duration_since_occupied_for_freespace_ms", C.c_int32),
        ("min_consecutive_occupancy_duration_for_reset_ms", C.c_int32),
        ("check_neighborhood", C.c_int32),
        ("initialize_to_high_confidence_freespace", C.c_int32),
        ("tsdf_weighting_variant", C.c_int32),
        ("tsdf_skip_at_negative_truncation", C.c_int32),
        ("tsdf_weight_clamp_before_blend", C.c_int32),
        ("color_occlusion_threshold_vox", C.c_float),
        ("esdf_propagation", C.c_int32),
        ("mesh_ambiguity_rule", C.c_int32),
        ("mesh_normal_rule", C.c_int32),
        ("decay_deallocate_decayed_blocks", C.c_int32),
        ("tsdf_set_free_distance_on_decayed", C.c_int32),
        ("tsdf_decayed_free_distance_vox", C.c_float),
        ("occupancy_decay_to_free", C.c_int32),
        ("slice_height_above_plane_m", C.c_float),
        ("slice_height_thickness_m", C.c_float),
        ("esdf_use_ground_plane", C.c_int32),
        ("esdf_ground_plane", C.c_float * 4),
    ]


class Camera(C.Structure):
    _fields_ = [("fu", C.c_float), ("fv", C.c_float), ("cu", C.c_float), ("cv", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Lidar(C.Structure):
    _fields_ = [("num_azimuth_divisions", C.c_int32), ("num_elevation_divisions", C.c_int32),
                ("min_valid_range_m", C.c_float), ("min_elevation_rad", C.c_float), ("max_elevation_rad", C.c_float)]


class BoundingShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3)]


class Index3D(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32)]


class DeviceView(C.Structure):
    _fields_ = [("table", C.c_void_p), ("table_mask", C.c_uint32), ("table_shift", C.c_uint32), ("slot_flags", C.c_void_p),
                ("slot_index", C.c_void_p), ("tsdf", C.c_void_p), ("color", C.c_void_p), ("esdf", C.c_void_p),
                ("voxel_size", C.c_float), ("block_capacity", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "blocks_allocated", "tsdf_blocks_in_view", "color_blocks_updated", "esdf_columns_marked", "esdf_blocks_swept",
        "esdf_window_voxels", "mesh_blocks_updated", "mesh_vertices", "mesh_triangles", "capacity_overflow", "lidar_blocks_beam_centric")]


# every symbol include/nvblox_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_pi32 = C.POINTER(C.c_int32)
SIGNATURES = {
    "nvbx_mapper_create": (C.c_int, [C.c_int, _vp, C.POINTER(Params), _i64, C.POINTER(_vp)]),
    "nvbx_mapper_destroy": (C.c_int, [_vp]),
    "nvbx_mapper_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "nvbx_mapper_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "nvbx_synchronize": (C.c_int, [_vp]),
    "nvbx_flush": (C.c_int, [_vp]),
    "nvbx_measure_depth": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera), _vp, _vp, _i64]),
    "nvbx_apply_measurements": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32]),
    "nvbx_mapper_set_max_capacity": (C.c_int, [_vp, _i64]),
    "nvbx_mapper_set_color_deferral": (C.c_int, [_vp, C.c_int32]),
    "nvbx_tsdf_zero_crossings": (_i64, [_vp, C.c_float, C.c_float, _vp, _i64]),
    "nvbx_fit_plane_ransac": (_i64, [_vp, _i64, C.c_float, C.c_int32, C.c_uint32, _vp]),
    "nvbx_mapper_capacity": (C.c_int64, [_vp]),
    "nvbx_integrate_depth_batch": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "nvbx_integrate_depth_pair": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "nvbx_integrate_color_batch": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "nvbx_take_cleared_blocks": (C.c_int64, [_vp, _vp, _i64]),
    "nvbx_get_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "nvbx_mapper_wait_for": (C.c_int, [_vp, _vp]),
    "nvbx_selftest_arith": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64]),
    "nvbx_decay_occupancy": (C.c_int, [_vp]),
    "nvbx_motion_compensate_pointcloud": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, C.c_float, _vp]),
    "nvbx_set_time_ms": (C.c_int, [_vp, C.c_int64]),
    "nvbx_detect_dynamics": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera), C.c_float, _vp]),
    "nvbx_remove_small_components": (C.c_int, [_vp, _vp, _i32, _i32, _i32]),
    "nvbx_dynamic_depth_split": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera), C.c_float, _i32, C.c_float, _vp, _vp, _vp, _vp]),
    "nvbx_set_view_export": (C.c_int, [_vp, _vp, _i64]),
    "nvbx_split_depth_by_mask": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, C.POINTER(Camera), C.POINTER(Camera), C.c_float, _vp, _vp, _vp]),
    "nvbx_split_color_by_mask": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "nvbx_default_params": (None, [C.POINTER(Params)]),
    "nvbx_backproject_depth": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(Camera), C.c_float, _vp, _i64, C.POINTER(_i64)]),
    "nvbx_transform_pointcloud": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "nvbx_esdf_slice_combined_size": (C.c_int, [_vp, _vp, C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "nvbx_esdf_slice_combined_to_image": (C.c_int, [_vp, _vp, C.c_float, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "nvbx_get_device_view": (C.c_int, [_vp, C.POINTER(DeviceView)]),
    "nvbx_last_error": (C.c_char_p, []),
    "nvbx_mapper_clear": (C.c_int, [_vp]),
    "nvbx_integrate_depth": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera)]),
    "nvbx_integrate_depth_u16mm": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera)]),
    "nvbx_integrate_color": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera)]),
    "nvbx_integrate_color_bgra8": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Camera)]),
    "nvbx_integrate_lidar_depth": (C.c_int, [_vp, _vp, _i32, _i32, _vp, C.POINTER(Lidar)]),
    "nvbx_depth_image_from_pointcloud": (C.c_int, [_vp, _vp, _i64, C.POINTER(Lidar), _vp]),
    "nvbx_update_esdf": (C.c_int, [_vp]),
    "nvbx_update_color_mesh": (C.c_int, [_vp, _i32]),
    "nvbx_decay_tsdf": (C.c_int, [_vp, _i32]),
    "nvbx_clear_outside_radius": (C.c_int, [_vp, _vp, _f]),
    "nvbx_clear_tsdf_inside_shapes": (C.c_int, [_vp, _vp, _i32]),
    "nvbx_esdf_slice_size": (C.c_int, [_vp, _pi32, _pi32, _vp]),
    "nvbx_esdf_slice_to_image": (C.c_int, [_vp, _f, _vp, _i64, _pi32, _pi32, _vp]),
    "nvbx_esdf_slice_to_host": (C.c_int, [_vp, _f, _vp, _i64, _pi32, _pi32, _vp]),
    "nvbx_occupancy_grid_from_slice": (C.c_int, [_vp, _vp, _i32, _i32, _f, _vp]),
    "nvbx_pointcloud_from_slice": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _f, _f, _vp, _pi32]),
    "nvbx_esdf_dense_grid": (C.c_int, [_vp, _vp, _vp, _f, _vp]),
    "nvbx_num_blocks": (_i64, [_vp, C.c_uint32]),
    "nvbx_block_indices": (_i64, [_vp, C.c_uint32, _vp, _i64]),
    "nvbx_get_block": (C.c_int, [_vp, C.c_uint32, Index3D, _vp]),
    "nvbx_set_block": (C.c_int, [_vp, C.c_uint32, Index3D, _vp]),
    "nvbx_get_blocks": (C.c_int, [_vp, C.c_uint32, _vp, _i64, _vp, _vp]),
    "nvbx_set_blocks": (C.c_int, [_vp, C.c_uint32, _vp, _i64, _vp]),
    "nvbx_save_map": (C.c_int, [_vp, C.c_char_p]),
    "nvbx_load_map": (C.c_int, [_vp, C.c_char_p]),
    "nvbx_last_depth_view": (_i64, [_vp, _vp, _i64]),
    "nvbx_last_color_view": (_i64, [_vp, _vp, _i64]),
    "nvbx_get_synthetic_depth": (C.c_int, [_vp, _vp, _i64, _pi32, _pi32]),
    "nvbx_get_counters": (C.c_int, [_vp, C.POINTER(Counters)]),
    "nvbx_mesh_sizes": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "nvbx_mesh_copy": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nvbx_esdf_dirty_list": (C.c_int, [_vp, _vp, _vp, _i64]),
    "nvbx_mark_esdf_dirty": (C.c_int, [_vp, _vp, _vp, _i64]),
    "nvbx_mark_esdf_dirty_gathered": (C.c_int, [_vp, _vp, _i32, _i32, _i64]),
    "nvbx_mark_esdf_dirty_gathered_deferred": (C.c_int, [_vp, _vp, _i32, _i32, _i64]),
    "nvbx_frame_acquire": (C.c_int, [C.c_int, C.c_size_t, _vp, C.POINTER(_vp)]),
    "nvbx_frame_retain": (C.c_int, [_vp]),
    "nvbx_frame_release": (C.c_int, [_vp]),
    "nvbx_frame_release_on": (C.c_int, [_vp, _vp]),
    "nvbx_frame_device": (_i32, [_vp]),
    "nvbx_frame_refcount": (_i32, [_vp]),
    "nvbx_frame_writable": (_i32, [_vp, _vp]),
    "nvbx_frame_pool_trim": (C.c_int, [C.c_int]),
    "nvbx_frame_pool_stats": (C.c_int, [C.POINTER(_i64)]),
    "nvbx_frame_upload": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "nvbx_color_image_acquire": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "nvbx_integrate_color_owned": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, C.POINTER(Camera)]),
    "nvbx_set_profiling": (C.c_int, [_vp, _i32]),
    "nvbx_get_profile": (C.c_int, [_vp, C.c_char_p, _i64]),
}

STREAM_UNKNOWN = C.c_void_p(-1)      # NVBX_STREAM_UNKNOWN

_lib = None


def load():
    """dlopen the HIP library.  torch is imported first so that both share ONE libamdhip64.so.7 (same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libnvblox_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C isaac_ros_nvblox_amd/csrc`. There is no CPU fallback for the product path." % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads torch's bundled HIP runtime first)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
