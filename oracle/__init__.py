"""ctypes binding of the CPU oracle (oracle/nvblox_oracle.c).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnvblox_oracle.so")

L_TSDF, L_COLOR, L_ESDF, L_MESH, L_FREESPACE = 1, 2, 4, 8, 32


class OrcParams(C.Structure):
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


def default_params(**kw):
    """fuser.yaml values (nvblox_examples_bringup/config/nvblox/fuser.yaml:24-42) + nvblox_base.yaml:87,103-107."""
    p = OrcParams(
        voxel_size=0.05, max_integration_distance_m=8.0, truncation_distance_vox=4.0, max_weight=5.0,
        weighting_mode=0, raycast_subsampling_factor=4,
        esdf_min_weight=0.1, esdf_max_site_distance_vox=2.0, esdf_max_distance_m=2.0,
        esdf_slice_height=0.09, esdf_slice_min_height=0.09, esdf_slice_max_height=0.65,
        mesh_min_weight=0.1, mesh_weld_vertices=1,
        sphere_tracing_subsampling=4, sphere_tracing_max_steps=100,
        sphere_tracing_max_ray_length_m=15.0, sphere_tracing_surface_eps_vox=0.1,
        tsdf_decay_factor=0.95, tsdf_decayed_weight_threshold=0.001,
        esdf_site_rule=0, depth_interp_nearest=0, lidar_max_integration_distance_m=10.0,
        lidar_linear_interpolation_max_allowable_difference_vox=2.0,
        lidar_nearest_interpolation_max_allowable_dist_to_ray_vox=0.5, invalid_depth_decay_factor=-1.0,
        projective_layer_type=0, free_region_occupancy_probability=0.45, occupied_region_occupancy_probability=0.55,
        unobserved_region_occupancy_probability=0.5, occupied_region_half_width_m=0.1,
        free_region_decay_probability=0.55, occupied_region_decay_probability=0.30, esdf_mode=0,
        max_tsdf_distance_for_occupancy_m=0.15, max_unobserved_to_keep_consecutive_occupancy_ms=200,
        min_duration_since_occupied_for_freespace_ms=1000, min_consecutive_occupancy_duration_for_reset_ms=2000,
        check_neighborhood=1, initialize_to_high_confidence_freespace=0,
        tsdf_weighting_variant=0, tsdf_skip_at_negative_truncation=0, tsdf_weight_clamp_before_blend=0,
        color_occlusion_threshold_vox=-1.0, esdf_propagation=0, mesh_ambiguity_rule=0, mesh_normal_rule=0,
        decay_deallocate_decayed_blocks=1, tsdf_set_free_distance_on_decayed=0, tsdf_decayed_free_distance_vox=4.0, occupancy_decay_to_free=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build(force=False):
    src = os.path.join(_HERE, "nvblox_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
        L.orc_create.restype = vp; L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_set_params.argtypes = [vp, C.POINTER(OrcParams)]
        L.orc_destroy.argtypes = [vp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_traversal_accumulate.argtypes = [C.c_int]; L.orc_set_traversal_accumulate.restype = None
        L.orc_index_hash.restype = C.c_uint32; L.orc_index_hash.argtypes = [i32, i32, i32]
        L.orc_integrate_depth.restype = i64; L.orc_integrate_depth.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_integrate_lidar_depth.restype = i64; L.orc_integrate_lidar_depth.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_depth_image_from_pointcloud.argtypes = [vp, i64, vp, vp]
        L.orc_split_depth_by_mask.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_float, vp, vp]
        L.orc_lidar_project.restype = C.c_int; L.orc_lidar_project.argtypes = [vp, vp, f32p, f32p]
        L.orc_atan2f.restype = C.c_float; L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_tsdf_zero_crossings.restype = i64; L.orc_tsdf_zero_crossings.argtypes = [vp, C.c_float, C.c_float, vp, i64]
        L.orc_fit_plane_ransac.restype = i64; L.orc_fit_plane_ransac.argtypes = [vp, i64, C.c_float, C.c_int32, C.c_uint32, vp]
        L.orc_lidar_sample_points.restype = None; L.orc_lidar_sample_points.argtypes = [C.POINTER(OrcParams), vp, vp, vp, i64, C.c_float, vp, vp]
        L.orc_integrate_color.restype = i64; L.orc_integrate_color.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_num_blocks.restype = i64; L.orc_num_blocks.argtypes = [vp, C.c_uint32]
        L.orc_block_indices.restype = i64; L.orc_block_indices.argtypes = [vp, C.c_uint32, vp, i64]
        L.orc_last_view.restype = i64; L.orc_last_view.argtypes = [vp, vp, i64]
        L.orc_last_color_view.restype = i64; L.orc_last_color_view.argtypes = [vp, vp, i64]
        L.orc_get_block.restype = C.c_int; L.orc_get_block.argtypes = [vp, C.c_uint32, i32, i32, i32, vp]
        L.orc_set_block.restype = C.c_int; L.orc_set_block.argtypes = [vp, C.c_uint32, i32, i32, i32, vp]
        L.orc_get_synthetic_depth.restype = C.c_int; L.orc_get_synthetic_depth.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_update_esdf.restype = i64; L.orc_update_esdf.argtypes = [vp]
        L.orc_esdf_slice_image.restype = i64
        L.orc_esdf_slice_image.argtypes = [vp, C.c_float, vp, i64, C.POINTER(i32), C.POINTER(i32), vp]
        L.orc_esdf_dense_grid.argtypes = [vp, vp, vp, C.c_float, vp]
        L.orc_update_mesh.restype = i64; L.orc_update_mesh.argtypes = [vp, C.c_int]
        L.orc_mesh_counts.restype = C.c_int; L.orc_mesh_counts.argtypes = [vp, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.orc_mesh_get.restype = C.c_int; L.orc_mesh_get.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        L.orc_decay_tsdf.restype = i64; L.orc_decay_tsdf.argtypes = [vp, C.c_int]
        L.orc_decay_occupancy.restype = i64; L.orc_decay_occupancy.argtypes = [vp]
        L.orc_set_time_ms.argtypes = [vp, C.c_int64]
        L.orc_motion_compensate_pointcloud.argtypes = [vp, vp, i64, vp, vp, C.c_float, vp]
        L.orc_detect_dynamics.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_float, vp]
        L.orc_remove_small_components.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
        L.orc_clear_tsdf_inside_shapes.restype = i64; L.orc_clear_tsdf_inside_shapes.argtypes = [vp, vp, i32]
        L.orc_clear_outside_radius.restype = i64; L.orc_clear_outside_radius.argtypes = [vp, vp, C.c_float]
        L.orc_measure_depth.restype = i64; L.orc_measure_depth.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, i64]
        L.orc_apply_measurements.restype = i64; L.orc_apply_measurements.argtypes = [vp, vp, vp, i32, i64, i32, i32]
        L.orc_take_cleared_blocks.restype = i64; L.orc_take_cleared_blocks.argtypes = [vp, vp, i64]
        L.orc_mark_esdf_dirty.restype = i64; L.orc_mark_esdf_dirty.argtypes = [vp, vp, i64]
        L.orc_esdf_dirty_list.restype = i64; L.orc_esdf_dirty_list.argtypes = [vp, vp, i64]
        _lib = L
    return _lib


TSDF_DT = np.dtype([("distance", "<f4"), ("weight", "<f4")])
COLOR_DT = np.dtype([("r", "u1"), ("g", "u1"), ("b", "u1"), ("pad", "u1"), ("weight", "<f4")])
ESDF_DT = np.dtype([("squared_distance_vox", "<f4"), ("parent_direction", "<i4", (3,)),
                    ("is_inside", "u1"), ("observed", "u1"), ("is_site", "u1"), ("pad", "u1")])
FREESPACE_DT = np.dtype([("last_occupied_timestamp_ms", "<i8"), ("consecutive_occupancy_duration_ms", "<i8"),
                         ("is_high_confidence_freespace", "u1"), ("initialized", "u1"), ("pad", "u1", (6,))])
_DT = {L_TSDF: TSDF_DT, L_COLOR: COLOR_DT, L_ESDF: ESDF_DT, L_FREESPACE: FREESPACE_DT}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleMap:
    """CPU oracle of nvblox::Mapper for the hot path (same method names as the C-ABI host mirror)."""

    def __init__(self, params=None):
        self.params = params or default_params()
        self._h = lib().orc_create(C.byref(self.params))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h); self._h = None

    def set_params(self, params):
        self.params = params
        lib().orc_set_params(self._h, C.byref(params))

    @staticmethod
    def _T(T):
        return np.ascontiguousarray(np.asarray(T, np.float32).reshape(4, 4))

    @staticmethod
    def _cam(cam):
        return np.ascontiguousarray(np.asarray(cam, np.float32).reshape(6))

    def integrate_depth(self, depth, T_L_C, cam):
        depth = np.ascontiguousarray(depth, np.float32)
        T = self._T(T_L_C); k = self._cam(cam)
        return lib().orc_integrate_depth(self._h, _p(depth), depth.shape[0], depth.shape[1], _p(T), _p(k))

    def integrate_lidar_depth(self, range_image, T_L_C, lidar):
        d = np.ascontiguousarray(range_image, np.float32); T = self._T(T_L_C)
        l5 = np.asarray(lidar, np.float32)
        return lib().orc_integrate_lidar_depth(self._h, _p(d), d.shape[0], d.shape[1], _p(T), _p(l5))

    def integrate_color(self, rgb, T_L_C, cam):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        T = self._T(T_L_C); k = self._cam(cam)
        return lib().orc_integrate_color(self._h, _p(rgb), rgb.shape[0], rgb.shape[1], _p(T), _p(k))

    def num_blocks(self, layer=L_TSDF):
        return lib().orc_num_blocks(self._h, layer)

    def block_indices(self, layer=L_TSDF):
        n = self.num_blocks(layer)
        out = np.zeros((max(n, 1), 3), np.int32)
        lib().orc_block_indices(self._h, layer, _p(out), n)
        return out[:n]

    def last_view(self):
        out = np.zeros((1 << 20, 3), np.int32)
        n = lib().orc_last_view(self._h, _p(out), out.shape[0])
        return out[:n].copy()

    def last_color_view(self):
        out = np.zeros((1 << 20, 3), np.int32)
        n = lib().orc_last_color_view(self._h, _p(out), out.shape[0])
        return out[:n].copy()

    def get_block(self, layer, idx):
        out = np.zeros(512, _DT[layer])
        ok = lib().orc_get_block(self._h, layer, int(idx[0]), int(idx[1]), int(idx[2]), _p(out))
        return out if ok else None

    def set_block(self, layer, idx, data):
        data = np.ascontiguousarray(data, _DT[layer]); assert data.size == 512
        return lib().orc_set_block(self._h, layer, int(idx[0]), int(idx[1]), int(idx[2]), _p(data))

    def synthetic_depth(self):
        r, c = C.c_int(), C.c_int()
        if not lib().orc_get_synthetic_depth(self._h, None, C.byref(r), C.byref(c)):
            return None
        out = np.zeros((r.value, c.value), np.float32)
        lib().orc_get_synthetic_depth(self._h, _p(out), C.byref(r), C.byref(c))
        return out

    def update_esdf(self):
        return lib().orc_update_esdf(self._h)

    def esdf_slice_image(self, unknown_value=1000.0):
        r, c = C.c_int32(), C.c_int32()
        aabb = np.zeros(6, np.float32)
        n = lib().orc_esdf_slice_image(self._h, unknown_value, None, 0, C.byref(r), C.byref(c), _p(aabb))
        img = np.zeros((r.value, c.value), np.float32)
        if n:
            lib().orc_esdf_slice_image(self._h, unknown_value, _p(img), n, C.byref(r), C.byref(c), _p(aabb))
        return img, aabb

    def esdf_dense_grid(self, min_vox, size_vox, default_value):
        mn = np.asarray(min_vox, np.int32); sz = np.asarray(size_vox, np.int32)
        out = np.zeros(tuple(int(s) for s in sz), np.float32)
        lib().orc_esdf_dense_grid(self._h, _p(mn), _p(sz), default_value, _p(out))
        return out

    def update_mesh(self, full=False):
        return lib().orc_update_mesh(self._h, int(full))

    def mesh_block(self, idx):
        nv, nt = C.c_int32(), C.c_int32()
        if not lib().orc_mesh_counts(self._h, int(idx[0]), int(idx[1]), int(idx[2]), C.byref(nv), C.byref(nt)):
            return None
        v = np.zeros((nv.value, 3), np.float32); n = np.zeros((nv.value, 3), np.float32)
        c = np.zeros((nv.value, 4), np.uint8); t = np.zeros((nt.value, 3), np.int32)
        lib().orc_mesh_get(self._h, int(idx[0]), int(idx[1]), int(idx[2]), _p(v), _p(n), _p(c), _p(t))
        return dict(vertices=v, normals=n, colors=c, triangles=t)

    def mark_esdf_dirty(self, indices):
        idx = np.ascontiguousarray(np.asarray(indices, np.int32).reshape(-1, 3))
        return lib().orc_mark_esdf_dirty(self._h, _p(idx), idx.shape[0])

    def esdf_dirty_list(self):
        out = np.zeros((1 << 18, 3), np.int32)
        n = lib().orc_esdf_dirty_list(self._h, _p(out), out.shape[0])
        return out[:n].copy()

    def decay_occupancy(self):
        return lib().orc_decay_occupancy(self._h)

    def set_time_ms(self, t):
        lib().orc_set_time_ms(self._h, int(t))

    def detect_dynamics(self, depth, T_L_C, cam, max_distance_m=0.0):
        d = np.ascontiguousarray(depth, np.float32); mask = np.zeros(d.shape, np.uint8)
        lib().orc_detect_dynamics(self._h, _p(d), d.shape[0], d.shape[1], _p(self._T(T_L_C)), _p(self._cam(cam)), float(max_distance_m), _p(mask))
        return mask

    def decay_tsdf(self, exclude_last_view=True):
        return lib().orc_decay_tsdf(self._h, int(exclude_last_view))

    def clear_tsdf_inside_shapes(self, shapes):
        rows = []
        for sh in shapes:
            rows.append([0.0] + [float(v) for v in sh[1]] + [float(sh[2]), 0.0, 0.0] if sh[0] == "sphere"
                        else [1.0] + [float(v) for v in sh[1]] + [float(v) for v in sh[2]])
        a = np.ascontiguousarray(np.asarray(rows, np.float32).reshape(-1, 7))
        return lib().orc_clear_tsdf_inside_shapes(self._h, _p(a), a.shape[0])

    def clear_outside_radius(self, center, radius):
        c = np.asarray(center, np.float32)
        return lib().orc_clear_outside_radius(self._h, _p(c), float(radius))

    # -- the CPU counterpart of Mapper.measure_depth / apply_measurements (torch CPU tensors or numpy arrays; dist.MeasurementFusion)
    def measure_depth(self, depth, T_L_C, cam, out_blocks, out_count):
        depth = np.ascontiguousarray(depth, np.float32); T = self._T(T_L_C); k = self._cam(cam)
        ob = out_blocks.numpy() if hasattr(out_blocks, "numpy") else out_blocks
        n = lib().orc_measure_depth(self._h, _p(depth), depth.shape[0], depth.shape[1], _p(T), _p(k), _p(ob), ob.shape[0])
        out_count[0] = int(n)

    def apply_measurements(self, gathered, counts, owner_mod=0, owner_rank=0):
        g = gathered.numpy() if hasattr(gathered, "numpy") else gathered
        c = np.ascontiguousarray(counts.numpy() if hasattr(counts, "numpy") else counts, np.int32)
        return lib().orc_apply_measurements(self._h, _p(g), _p(c), g.shape[0], g.shape[1], int(owner_mod), int(owner_rank))

    def tsdf_zero_crossings(self, min_z_m, max_z_m):
        out = np.zeros((1 << 20, 3), np.float32)
        n = lib().orc_tsdf_zero_crossings(self._h, C.c_float(min_z_m), C.c_float(max_z_m), _p(out), out.shape[0])
        assert n <= out.shape[0]
        return out[:n].copy()

    def take_cleared_blocks(self):
        out = np.zeros((1 << 16, 3), np.int32)
        n = lib().orc_take_cleared_blocks(self._h, _p(out), out.shape[0])
        return out[:n].copy()


def set_traversal_accumulate(on):
    """1: the view calculation walks with the textbook accumulated crossing parameters instead of the closed form (a cross-check, process-wide)."""
    lib().orc_set_traversal_accumulate(int(bool(on)))


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


def depth_image_from_pointcloud(points, lidar):
    pts = np.ascontiguousarray(points, np.float32); l5 = np.asarray(lidar, np.float32)
    img = np.zeros((int(lidar[1]), int(lidar[0])), np.float32)
    lib().orc_depth_image_from_pointcloud(_p(pts), pts.shape[0], _p(l5), _p(img))
    return img


def motion_compensate_pointcloud(points, rel_time_ms, T_L_S_start, T_L_S_end, scan_duration_ms):
    p = np.ascontiguousarray(points, np.float32); t = np.ascontiguousarray(rel_time_ms, np.float32)
    T0 = np.ascontiguousarray(np.asarray(T_L_S_start, np.float32).reshape(4, 4)); T1 = np.ascontiguousarray(np.asarray(T_L_S_end, np.float32).reshape(4, 4))
    out = np.zeros_like(p)
    lib().orc_motion_compensate_pointcloud(_p(p), _p(t), p.shape[0], _p(T0), _p(T1), float(scan_duration_ms), _p(out))
    return out


def remove_small_components(mask, min_size):
    mk = np.ascontiguousarray(mask, np.uint8).copy()
    lib().orc_remove_small_components(_p(mk), mk.shape[0], mk.shape[1], int(min_size))
    return mk


def split_depth_by_mask(depth, mask, T_CM_CD, depth_cam, mask_cam, occlusion_threshold_m):
    d = np.ascontiguousarray(depth, np.float32); mk = np.ascontiguousarray(mask, np.uint8)
    T = np.ascontiguousarray(np.asarray(T_CM_CD, np.float32).reshape(4, 4))
    dc = np.asarray(depth_cam, np.float32); mc = np.asarray(mask_cam, np.float32)
    un = np.zeros_like(d); ma = np.zeros_like(d)
    lib().orc_split_depth_by_mask(_p(d), d.shape[0], d.shape[1], _p(mk), mk.shape[0], mk.shape[1], _p(T), _p(dc), _p(mc),
                                  float(occlusion_threshold_m), _p(un), _p(ma))
    return un, ma


def fit_plane_ransac(points, distance_threshold_m, iterations, seed=1):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3); plane = np.zeros(4, np.float32)
    n = lib().orc_fit_plane_ransac(_p(p), C.c_int64(len(p)), C.c_float(distance_threshold_m), C.c_int32(iterations), C.c_uint32(seed), _p(plane))
    return plane, int(n)


def lidar_sample_points(params, lidar, range_image, pts, max_dist):
    """The oracle's LiDAR measurement model at sensor-frame points [N, 3] -> (branch int32 [N], ds float32 [N]); tests only."""
    l5 = np.asarray(lidar, np.float32); img = np.ascontiguousarray(range_image, np.float32); q = np.ascontiguousarray(pts, np.float32)
    br = np.zeros(len(q), np.int32); ds = np.zeros(len(q), np.float32)
    lib().orc_lidar_sample_points(C.byref(params), _p(l5), _p(img), _p(q), C.c_int64(len(q)), C.c_float(max_dist), _p(br), _p(ds))
    return br, ds


def lidar_project(lidar, p):
    l5 = np.asarray(lidar, np.float32); q = np.asarray(p, np.float32)
    u, v = C.c_float(), C.c_float()
    ok = lib().orc_lidar_project(_p(l5), _p(q), C.byref(u), C.byref(v))
    return (u.value, v.value) if ok else None
