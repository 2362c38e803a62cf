"""Host-side mirror of nvblox::Mapper / MultiMapper / EsdfSlicer for tests and bench.py (thin ctypes over the C-ABI).

Method names follow the reference call sites (nvblox_ros/src/lib/nvblox_node.cpp:781,1062,1264;
layer_publishing.cpp:686-689).  Images are torch CUDA tensors (device memory, like the reference's
MemoryType::kDevice images) or numpy arrays (uploaded first).
"""
import ctypes as C
import os
import numpy as np

from . import _lib
from ._lib import BoundingShape, Camera, Counters, Index3D, Lidar, Params

LAYER_TSDF, LAYER_COLOR, LAYER_ESDF, LAYER_MESH, LAYER_OCCUPANCY, LAYER_FREESPACE = 1, 2, 4, 8, 16, 32

TSDF_DT = np.dtype([("distance", "<f4"), ("weight", "<f4")])
COLOR_DT = np.dtype([("r", "u1"), ("g", "u1"), ("b", "u1"), ("pad", "u1"), ("weight", "<f4")])
ESDF_DT = np.dtype([("squared_distance_vox", "<f4"), ("parent_direction", "<i4", (3,)),
                    ("is_inside", "u1"), ("observed", "u1"), ("is_site", "u1"), ("pad", "u1")])
OCCUPANCY_DT = np.dtype([("log_odds", "<f4")])
FREESPACE_DT = np.dtype([("last_occupied_timestamp_ms", "<i8"), ("consecutive_occupancy_duration_ms", "<i8"),
                         ("is_high_confidence_freespace", "u1"), ("initialized", "u1"), ("pad", "u1", (6,))])
_DT = {LAYER_TSDF: TSDF_DT, LAYER_COLOR: COLOR_DT, LAYER_ESDF: ESDF_DT, LAYER_OCCUPANCY: OCCUPANCY_DT, LAYER_FREESPACE: FREESPACE_DT}


def default_params(**kw):
    """fuser.yaml values (nvblox_examples_bringup/config/nvblox/fuser.yaml:24-42) + nvblox_base.yaml:87,103-107."""
    p = Params(
        voxel_size=0.05, max_integration_distance_m=8.0, truncation_distance_vox=4.0, max_weight=5.0,
        weighting_mode=0, raycast_subsampling_factor=4,
        esdf_min_weight=0.1, esdf_max_site_distance_vox=2.0, esdf_max_distance_m=2.0,
        esdf_slice_height=0.09, esdf_slice_min_height=0.09, esdf_slice_max_height=0.65,
        mesh_min_weight=0.1, mesh_weld_vertices=1,
        sphere_tracing_subsampling=4, sphere_tracing_max_steps=100,
        sphere_tracing_max_ray_length_m=15.0, sphere_tracing_surface_eps_vox=0.1,
        tsdf_decay_factor=0.95, tsdf_decayed_weight_threshold=0.001,
        esdf_site_rule=0, depth_interp_nearest=0,
        lidar_max_integration_distance_m=10.0,
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


class NvbxError(RuntimeError):
    pass


# NVBX_CHECK_ON_CLOSE=1 with the -DNVBX_CHECK_INVARIANTS variant of the library: what the mappers closed so far reported (tests/test_gpu_invariants.py)
INVARIANT_REPORT = {"mappers_checked": 0, "violations": []}


def frame_pool_stats():
    """(held, free, bytes, created, waits, syncs) of the library's frame pool (nvbx_frame_pool_stats)."""
    out = (C.c_int64 * 6)()
    _lib.load().nvbx_frame_pool_stats(out)
    return tuple(int(v) for v in out)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class ColorFrame:
    """A colour image in a library-owned, reference-counted device frame (nvbx_frame_acquire, include/nvblox_hip.h): what nvblox::Image<Color>
    device memory is in the C++ facade.  A mapper that holds integrate_color back RETAINS the frame instead of copying the image; `write` first makes
    sure nobody else holds the frame and continues in another one if a mapper does (rotation) -- the node's one re-used colour buffer
    (nvblox_node.hpp:484-488) without the staging copy."""

    def __init__(self, rows, cols, channels=3, device=0):
        self.lib = _lib.load()
        self.rows, self.cols, self.channels, self.device = int(rows), int(cols), int(channels), int(device)
        self.nbytes = self.rows * self.cols * self.channels
        self._p = C.c_void_p()
        self._acquire(_lib.STREAM_UNKNOWN)

    def _acquire(self, stream):
        p = C.c_void_p()
        if self.lib.nvbx_frame_acquire(self.device, self.nbytes, stream, C.byref(p)) != 0:
            raise NvbxError("nvbx_frame_acquire: %s" % self.lib.nvbx_last_error().decode())
        self._p = p

    @property
    def ptr(self):
        return self._p.value

    def shared(self):
        return self.lib.nvbx_frame_refcount(self._p) > 1

    def write(self, src, stream=None):
        """Overwrite the image from a torch CUDA tensor / numpy array of rows x cols x channels uint8: asynchronously on `stream` (a raw hipStream_t),
        or with a blocking copy (stream=None: a writer the library knows nothing about).  Continues in another frame first when a mapper still holds
        this one, or its launches may still be reading it."""
        s = C.c_void_p(stream) if stream else _lib.STREAM_UNKNOWN
        if self.lib.nvbx_frame_writable(self._p, s) == 0:      # a mapper holds it, or its launches may still be reading it (and not on this stream)
            old = self._p
            self._acquire(s)
            self.lib.nvbx_frame_release(old)
        if hasattr(src, "data_ptr"):
            assert src.is_contiguous() and src.numel() * src.element_size() == self.nbytes
            sp = C.c_void_p(src.data_ptr())
        else:
            a = np.ascontiguousarray(src, np.uint8); assert a.nbytes == self.nbytes
            self._host_keep = a; sp = _np_ptr(a)
        if self.lib.nvbx_frame_upload(self._p, sp, self.nbytes, s) != 0:
            raise NvbxError("nvbx_frame_upload: %s" % self.lib.nvbx_last_error().decode())
        return self

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.lib.nvbx_frame_release(self._p); self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mapper:
    def __init__(self, params=None, device=0, block_capacity=1 << 15, stream=None, max_block_capacity=None):
        import torch
        if not torch.cuda.is_available():
            raise NvbxError("no HIP device visible: the product path has no CPU fallback")
        self._torch = torch
        self.lib = _lib.load()
        self.params = params or default_params()
        self.device = device
        self._h = C.c_void_p()
        torch.cuda.set_device(device)
        s = C.c_void_p(stream) if stream else None
        self._check(self.lib.nvbx_mapper_create(device, s, C.byref(self.params), block_capacity, C.byref(self._h)))
        self._capacity0 = block_capacity
        self._keep = []
        if max_block_capacity is not None:
            self.set_max_capacity(max_block_capacity)

    @property
    def capacity(self):
        """Current block capacity of the HBM pools (they double on demand up to max_block_capacity)."""
        return int(self.lib.nvbx_mapper_capacity(self._h))

    def set_max_capacity(self, max_blocks):
        self._check(self.lib.nvbx_mapper_set_max_capacity(self._h, int(max_blocks)))

    def set_color_deferral(self, enable, staged=True):
        """Hold integrateColor (and an updateEsdf behind it) back until the next integrateDepth carries them out in pipelined order (two
        launches per frame instead of four; include/nvblox_hip.h nvbx_mapper_set_color_deferral).  A NEW mapper does so in the staged form
        (enable=True, staged=True: the mapper copies a held-back frame into its own memory first, nothing observable changes but the time).
        staged=False (opt-in, zero-copy): the colour image handed to integrate_color must then stay valid and unchanged until the next call into
        the mapper has returned.  enable=False: the classic order, every call launches its own kernels.
        (staged=True is the default here as in C and C++ -- ADVICE r04.)  An image in a ColorFrame is never copied, in either form."""
        self._check(self.lib.nvbx_mapper_set_color_deferral(self._h, (2 if staged else 1) if enable else 0))

    # -- ground plane (MultiMapper::ground_plane_estimator())
    def tsdf_zero_crossings(self, min_z_m, max_z_m):
        """xyz [N, 3] float32 of the TSDF's upward zero crossings with height in [min_z_m, max_z_m], sorted by (x, y, z)."""
        n = self._check(self.lib.nvbx_tsdf_zero_crossings(self._h, float(min_z_m), float(max_z_m), None, 0))
        out = np.zeros((max(n, 1), 3), np.float32)
        n2 = self._check(self.lib.nvbx_tsdf_zero_crossings(self._h, float(min_z_m), float(max_z_m), _np_ptr(out), out.shape[0]))
        return out[:min(n, n2)]

    def fit_plane_ransac(self, points, distance_threshold_m, iterations, seed=1):
        """-> (plane [nx, ny, nz, d] float32, inliers); host only"""
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 3); plane = np.zeros(4, np.float32)
        n = self._check(self.lib.nvbx_fit_plane_ransac(_np_ptr(p), len(p), float(distance_threshold_m), int(iterations), int(seed), _np_ptr(plane)))
        return plane, n

    def _check(self, rc):
        if rc < 0:
            raise NvbxError("nvbx error %d: %s" % (rc, self.lib.nvbx_last_error().decode()))
        return rc

    def invariant_violations(self, selftest=False):
        """-DNVBX_CHECK_INVARIANTS variant of the library only (NVBX_LIB=...; tools/build_variant.sh inv): (I1, I3, I4, writers still running, I8) as
        the kernels and the host counted them; None with the product library, which compiles the checks to nothing."""
        try:
            fn = self.lib.nvbx_debug_invariants
        except AttributeError:
            return None
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        out = (C.c_int64 * 5)()
        self._check(fn(self._h, out, int(bool(selftest))))
        return tuple(int(v) for v in out)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if os.environ.get("NVBX_CHECK_ON_CLOSE") == "1":        # (tests/test_gpu_invariants.py: every mapper of a test run answers for its launches)
                v = self.invariant_violations()
                if v is not None:
                    INVARIANT_REPORT["mappers_checked"] += 1
                    if any(v):          # (recorded as well as raised: most mappers are closed by __del__, which swallows exceptions)
                        INVARIANT_REPORT["violations"].append(v)
                        self.lib.nvbx_mapper_destroy(self._h); self._h = C.c_void_p()
                        raise NvbxError("invariant violations (I1, I3, I4, writers running, I8) = %r" % (v,))
            self.lib.nvbx_mapper_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _hold(self, slot, tensors):
        """Keep the device images of the call that has just been enqueued alive (`slot`: "_keep" depth / range images, "_keep_c" colour) and let go
        of the previous call's.  The kernels that read them run on the MAPPER's stream, which torch's caching allocator knows nothing about: a
        block it takes back could be handed out again and overwritten by the next upload (on torch's current stream) while kernels queued behind
        a slow launch -- a first launch loads its code object, ~0.5 ms -- still read it (found as garbage view rays in tests/test_gpu_pipeline.py).
        So before the old images are dropped, torch's current stream is made to wait for everything queued on the mapper's stream so far."""
        old = getattr(self, slot, None)
        if old:
            torch = self._torch
            cur = torch.cuda.current_stream(self.device); ms = self.torch_stream()
            if ms.cuda_stream != cur.cuda_stream:
                cur.wait_stream(ms)
        setattr(self, slot, tensors)

    def _dev(self, a, dtype):
        torch = self._torch
        if isinstance(a, torch.Tensor):
            t = a
            if not t.is_cuda:
                t = t.cuda(self.device)
            t = t.contiguous()
            assert t.dtype == dtype, (t.dtype, dtype)
            return t
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda(self.device)

    @staticmethod
    def _T(T):
        return np.ascontiguousarray(np.asarray(T, np.float32).reshape(4, 4))

    @staticmethod
    def _cam(cam):
        if isinstance(cam, Camera):
            return cam
        return Camera(float(cam[0]), float(cam[1]), float(cam[2]), float(cam[3]), int(cam[4]), int(cam[5]))

    def set_params(self, params):
        self._check(self.lib.nvbx_mapper_set_params(self._h, C.byref(params)))
        self.params = params

    def synchronize(self):
        self._check(self.lib.nvbx_synchronize(self._h))

    def wait_for(self, producer):
        """This mapper's stream waits for everything enqueued so far on `producer`'s stream (two mappers on two streams; nvbx_mapper_wait_for)."""
        self._check(self.lib.nvbx_mapper_wait_for(self._h, producer._h))

    def stream_handle(self):
        """The raw hipStream_t of the mapper (int)."""
        h = C.c_void_p()
        self._check(self.lib.nvbx_get_stream(self._h, C.byref(h)))
        return h.value or 0

    def torch_stream(self):
        """The mapper's stream as a torch stream object (for event ordering against work on torch's streams)."""
        if getattr(self, "_tstream", None) is None:
            self._tstream = self._torch.cuda.ExternalStream(self.stream_handle(), device="cuda:%d" % self.device)
        return self._tstream

    def flush(self):
        """Enqueue held-back work (the EDT of the last update_esdf) without waiting."""
        self._check(self.lib.nvbx_flush(self._h))

    def clear(self):
        self._check(self.lib.nvbx_mapper_clear(self._h))

    # -- integration (async)
    def integrate_depth(self, depth, T_L_C, cam):
        torch = self._torch
        if isinstance(depth, torch.Tensor) and depth.dtype == torch.int16 or (
                not isinstance(depth, torch.Tensor) and np.asarray(depth).dtype == np.uint16):
            if not isinstance(depth, torch.Tensor):
                depth = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16))
            d = self._dev(depth, torch.int16)
            fn = self.lib.nvbx_integrate_depth_u16mm
        else:
            d = self._dev(depth, torch.float32)
            fn = self.lib.nvbx_integrate_depth
        T = self._T(T_L_C); k = self._cam(cam)
        self._check(fn(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k)))
        self._hold("_keep", [d])   # keep the device image alive until the next call (stream-ordered use)

    @staticmethod
    def _lidar(lidar):
        if isinstance(lidar, Lidar):
            return lidar
        return Lidar(int(lidar[0]), int(lidar[1]), float(lidar[2]), float(lidar[3]), float(lidar[4]))

    def integrate_lidar_depth(self, range_image, T_L_C, lidar):
        """range image [elevation divisions, azimuth divisions] f32 metres; lidar = (cols, rows, min_range, min_el, max_el)."""
        d = self._dev(range_image, self._torch.float32)
        T = self._T(T_L_C); k = self._lidar(lidar)
        self._check(self.lib.nvbx_integrate_lidar_depth(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k)))
        self._hold("_keep", [d])

    def prepare_lidar(self, range_image, T_L_C, lidar):
        d = self._dev(range_image, self._torch.float32)
        T = self._T(T_L_C); k = self._lidar(lidar)
        return (self.lib.nvbx_integrate_lidar_depth, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k), (d, T, k))

    def motion_compensate_pointcloud(self, points, rel_time_ms, T_L_S_start, T_L_S_end, scan_duration_ms):
        """LiDAR motion compensation: points [n, 3] measured at rel_time_ms [n] within the scan -> sensor frame at scan start (device tensor)."""
        torch = self._torch
        p = self._dev(np.ascontiguousarray(points, np.float32) if not isinstance(points, torch.Tensor) else points, torch.float32)
        t = self._dev(np.ascontiguousarray(rel_time_ms, np.float32) if not isinstance(rel_time_ms, torch.Tensor) else rel_time_ms, torch.float32)
        out = torch.empty_like(p)
        self._check(self.lib.nvbx_motion_compensate_pointcloud(self._h, C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), p.shape[0],
                                                               _np_ptr(self._T(T_L_S_start)), _np_ptr(self._T(T_L_S_end)), float(scan_duration_ms),
                                                               C.c_void_p(out.data_ptr())))
        self.synchronize()
        return out

    def depth_image_from_pointcloud(self, points, lidar):
        torch = self._torch
        p = self._dev(points, torch.float32)
        assert p.dim() == 2 and p.shape[1] == 3
        k = self._lidar(lidar)
        img = torch.zeros((k.num_elevation_divisions, k.num_azimuth_divisions), dtype=torch.float32, device=p.device)
        self._check(self.lib.nvbx_depth_image_from_pointcloud(self._h, C.c_void_p(p.data_ptr()), p.shape[0], C.byref(k), C.c_void_p(img.data_ptr())))
        self.synchronize()
        return img

    def integrate_color(self, rgb, T_L_C, cam):
        """rgb8 [rows, cols, 3] or bgra8 [rows, cols, 4] (the two encodings image_conversions.cpp:170-176 accepts)."""
        if isinstance(rgb, ColorFrame):
            T = self._T(T_L_C); k = self._cam(cam)
            fn = self.lib.nvbx_integrate_color_bgra8 if rgb.channels == 4 else self.lib.nvbx_integrate_color
            self._check(fn(self._h, C.c_void_p(rgb.ptr), rgb.rows, rgb.cols, _np_ptr(T), C.byref(k)))
            return
        d = self._dev(rgb, self._torch.uint8)
        assert d.dim() == 3 and d.shape[2] in (3, 4)
        T = self._T(T_L_C); k = self._cam(cam)
        if d.shape[2] == 4:
            self._check(self.lib.nvbx_integrate_color_bgra8(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k)))
            self._hold("_keep_c", [d])
            return
        self._check(self.lib.nvbx_integrate_color(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k)))
        self._hold("_keep_c", [d])

    # -- pre-marshalled calls: the reference hands ready C++ objects to integrateDepth/Color; this keeps Python's per-call
    #    argument marshalling (tensor checks, numpy pose copy, struct construction) out of a measured loop
    def prepare_depth(self, depth, T_L_C, cam):
        torch = self._torch
        d = self._dev(depth, torch.float32)
        T = self._T(T_L_C); k = self._cam(cam)
        return (self.lib.nvbx_integrate_depth, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k), (d, T, k))

    def prepare_color(self, rgb, T_L_C, cam):
        if isinstance(rgb, ColorFrame):
            assert rgb.channels == 3
            T = self._T(T_L_C); k = self._cam(cam)
            return (self.lib.nvbx_integrate_color, C.c_void_p(rgb.ptr), rgb.rows, rgb.cols, _np_ptr(T), C.byref(k), (rgb, T, k))
        d = self._dev(rgb, self._torch.uint8)
        assert d.dim() == 3 and d.shape[2] == 3
        T = self._T(T_L_C); k = self._cam(cam)
        return (self.lib.nvbx_integrate_color, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k), (d, T, k))

    # -- camera batches (nvbx_integrate_depth_batch / _color_batch): n frames of one image size in ONE launch set
    def _batch(self, fn, imgs, dtype, poses, cams):
        n = len(imgs)
        d = [i if isinstance(i, ColorFrame) else self._dev(i, dtype) for i in imgs]
        ptrs = (C.c_void_p * n)(*[t.ptr if isinstance(t, ColorFrame) else t.data_ptr() for t in d])
        T = np.ascontiguousarray(np.stack([self._T(p) for p in poses]).reshape(n, 16))
        if not isinstance(cams, (list, tuple)) or not isinstance(cams[0], (list, tuple, Camera)):
            cams = [cams] * n
        ks = (Camera * n)(*[self._cam(k) for k in cams])
        r_, c_ = (d[0].rows, d[0].cols) if isinstance(d[0], ColorFrame) else (d[0].shape[0], d[0].shape[1])
        return (fn, n, ptrs, r_, c_, _np_ptr(T), ks, (d, T))

    def prepare_depth_batch(self, depths, poses, cams):
        return self._batch(self.lib.nvbx_integrate_depth_batch, depths, self._torch.float32, poses, cams)

    def prepare_color_batch(self, rgbs, poses, cams):
        return self._batch(self.lib.nvbx_integrate_color_batch, rgbs, self._torch.uint8, poses, cams)

    def integrate_prepared_batch(self, a):
        rc = a[0](self._h, a[1], a[2], a[3], a[4], a[5], a[6])
        if rc < 0:
            self._check(rc)

    def integrate_depth_batch(self, depths, poses, cams):
        a = self.prepare_depth_batch(depths, poses, cams); self.integrate_prepared_batch(a); self._hold("_keep", [a])

    def integrate_color_batch(self, rgbs, poses, cams):
        a = self.prepare_color_batch(rgbs, poses, cams); self.integrate_prepared_batch(a); self._hold("_keep_c", [a])

    def integrate_depth_pair(self, depth_a, other, depth_b, T_L_C, cam):
        """nvbx_integrate_depth_pair: self.integrate_depth(depth_a) followed by other.integrate_depth(depth_b) -- the background / foreground mappers of a
        MultiMapper's dynamic and human mapping types, fed the two halves of one mask-split depth frame -- in two launches instead of four (both mappers on
        one stream; anything else falls back to the two calls).  Same maps, bit for bit."""
        da = self._dev(depth_a, self._torch.float32); db = self._dev(depth_b, self._torch.float32)
        assert da.shape == db.shape
        self._check(self.lib.nvbx_integrate_depth_pair(self._h, C.c_void_p(da.data_ptr()), other._h, C.c_void_p(db.data_ptr()), da.shape[0], da.shape[1],
                                                       _np_ptr(self._T(T_L_C)), C.byref(self._cam(cam))))
        self._hold("_keep", [da, db]); other._hold("_keep", [db])

    def integrate_prepared(self, a):
        rc = a[0](self._h, a[1], a[2], a[3], a[4], a[5])
        if rc < 0:
            self._check(rc)

    def update_esdf(self):
        self._check(self.lib.nvbx_update_esdf(self._h))

    def update_color_mesh(self, full=False):
        self._check(self.lib.nvbx_update_color_mesh(self._h, int(full)))

    update_mesh = update_color_mesh

    def decay_tsdf(self, exclude_last_view=True):
        self._check(self.lib.nvbx_decay_tsdf(self._h, int(exclude_last_view)))

    def set_time_ms(self, t):
        """update_time_ms of the next integrate_depth (freespace layer of a projective_layer_type 2 mapper)."""
        self._check(self.lib.nvbx_set_time_ms(self._h, int(t)))

    def detect_dynamics(self, depth, T_L_C, cam, max_distance_m=0.0):
        """DynamicsDetection::computeDynamics: uint8 device tensor [rows, cols], 1 = the pixel's point lies in high-confidence freespace."""
        torch = self._torch
        d = self._dev(depth, torch.float32)
        mask = torch.empty(d.shape, dtype=torch.uint8, device=d.device)
        self._check(self.lib.nvbx_detect_dynamics(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(self._T(T_L_C)),
                                                  C.byref(self._cam(cam)), float(max_distance_m), C.c_void_p(mask.data_ptr())))
        self.synchronize()
        return mask

    # -- no-sync forms for a stream-ordered caller (outputs pre-allocated on the mapper's device; the caller's tensors must live on
    #    the mapper's stream or be ordered against it)
    def detect_dynamics_into(self, depth_dev, T_L_C, cam, max_distance_m, mask_out):
        self._check(self.lib.nvbx_detect_dynamics(self._h, C.c_void_p(depth_dev.data_ptr()), depth_dev.shape[0], depth_dev.shape[1], _np_ptr(self._T(T_L_C)),
                                                  C.byref(self._cam(cam)), float(max_distance_m), C.c_void_p(mask_out.data_ptr())))

    def remove_small_components_inplace(self, mask_dev, min_size):
        """(asynchronous on the mapper's stream)"""
        self._check(self.lib.nvbx_remove_small_components(self._h, C.c_void_p(mask_dev.data_ptr()), mask_dev.shape[0], mask_dev.shape[1], int(min_size)))

    def split_depth_by_mask_into(self, depth_dev, mask_dev, T_CM_CD, depth_cam, mask_cam, occlusion_threshold_m, unmasked_out, masked_out):
        self._check(self.lib.nvbx_split_depth_by_mask(self._h, C.c_void_p(depth_dev.data_ptr()), depth_dev.shape[0], depth_dev.shape[1],
                                                      C.c_void_p(mask_dev.data_ptr()), mask_dev.shape[0], mask_dev.shape[1], _np_ptr(self._T(T_CM_CD)),
                                                      C.byref(self._cam(depth_cam)), C.byref(self._cam(mask_cam)), float(occlusion_threshold_m),
                                                      C.c_void_p(unmasked_out.data_ptr()), C.c_void_p(masked_out.data_ptr()), None))

    def dynamic_depth_split_into(self, depth_dev, T_L_C, cam, max_distance_m, min_component_size, occlusion_threshold_m, mask_out, unmasked_out, masked_out, overlay_out=None):
        """The dynamic-mapping frame's front end in one call (three launches): detect_dynamics -> remove_small_components -> split_depth_by_mask."""
        self._check(self.lib.nvbx_dynamic_depth_split(self._h, C.c_void_p(depth_dev.data_ptr()), depth_dev.shape[0], depth_dev.shape[1], _np_ptr(self._T(T_L_C)),
                                                      C.byref(self._cam(cam)), float(max_distance_m), int(min_component_size), float(occlusion_threshold_m),
                                                      C.c_void_p(mask_out.data_ptr()), C.c_void_p(unmasked_out.data_ptr()), C.c_void_p(masked_out.data_ptr()),
                                                      C.c_void_p(overlay_out.data_ptr()) if overlay_out is not None else None))

    def remove_small_components(self, mask, min_size):
        torch = self._torch
        mk = self._dev(mask, torch.uint8).clone()
        torch.cuda.current_stream(mk.device).synchronize()      # the clone ran on torch's stream, the library works on the mapper's
        self._check(self.lib.nvbx_remove_small_components(self._h, C.c_void_p(mk.data_ptr()), mk.shape[0], mk.shape[1], int(min_size)))
        self.synchronize()          # (the call is asynchronous on the mapper's stream; the tensor goes back to torch's)
        return mk

    def decay_occupancy(self):
        """Mapper::decayOccupancyAllVoxels (nvblox_node.cpp:925-929); occupancy mappers only."""
        self._check(self.lib.nvbx_decay_occupancy(self._h))

    def clear_outside_radius(self, center, radius):
        c = np.asarray(center, np.float32)
        self._check(self.lib.nvbx_clear_outside_radius(self._h, _np_ptr(c), float(radius)))

    def take_cleared_blocks(self):
        """Mapper::getClearedBlocks (layer_publishing.cpp:716): blocks deallocated by decay / radius clearing since the last call."""
        out = np.zeros((self.capacity, 3), np.int32)
        n = self._check(self.lib.nvbx_take_cleared_blocks(self._h, _np_ptr(out), out.shape[0]))
        return out[:n].copy()

    def clear_tsdf_inside_shapes(self, shapes):
        """shapes: list of ("sphere", centre, radius) / ("aabb", min_corner, max_corner)."""
        arr = (BoundingShape * max(1, len(shapes)))()
        for i, sh in enumerate(shapes):
            if sh[0] == "sphere":
                arr[i].kind = 0; arr[i].a[:] = [float(v) for v in sh[1]]; arr[i].b[:] = [float(sh[2]), 0.0, 0.0]
            else:
                arr[i].kind = 1; arr[i].a[:] = [float(v) for v in sh[1]]; arr[i].b[:] = [float(v) for v in sh[2]]
        self._check(self.lib.nvbx_clear_tsdf_inside_shapes(self._h, C.cast(arr, C.c_void_p), len(shapes)))

    # -- queries (synchronise)
    def counters(self):
        c = Counters()
        self._check(self.lib.nvbx_get_counters(self._h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def num_blocks(self, layer=LAYER_TSDF):
        return self._check(self.lib.nvbx_num_blocks(self._h, layer))

    def block_indices(self, layer=LAYER_TSDF):
        n = self.num_blocks(layer)
        out = np.zeros((max(n, 1), 3), np.int32)
        n2 = self._check(self.lib.nvbx_block_indices(self._h, layer, _np_ptr(out), out.shape[0]))
        return out[:min(n, n2)]

    def last_view(self):
        out = np.zeros((self.capacity, 3), np.int32)
        n = self._check(self.lib.nvbx_last_depth_view(self._h, _np_ptr(out), out.shape[0]))
        return out[:n].copy()

    def last_color_view(self):
        out = np.zeros((self.capacity, 3), np.int32)
        n = self._check(self.lib.nvbx_last_color_view(self._h, _np_ptr(out), out.shape[0]))
        return out[:n].copy()

    def get_blocks(self, layer, indices):
        idx = np.ascontiguousarray(np.asarray(indices, np.int32).reshape(-1, 3))
        out = np.zeros((idx.shape[0], 512), _DT[layer])
        found = np.zeros(idx.shape[0], np.int32)
        self._check(self.lib.nvbx_get_blocks(self._h, layer, _np_ptr(idx), idx.shape[0], _np_ptr(out), _np_ptr(found)))
        return out, found.astype(bool)

    def get_block(self, layer, idx):
        out, found = self.get_blocks(layer, [idx])
        return out[0] if found[0] else None

    def set_block(self, layer, idx, data):
        data = np.ascontiguousarray(data, _DT[layer]); assert data.size == 512
        self._check(self.lib.nvbx_set_block(self._h, layer, Index3D(int(idx[0]), int(idx[1]), int(idx[2])), _np_ptr(data)))

    def set_blocks(self, layer, indices, data):
        idx = np.ascontiguousarray(np.asarray(indices, np.int32).reshape(-1, 3))
        data = np.ascontiguousarray(data, _DT[layer]); assert data.size == 512 * idx.shape[0]
        self._check(self.lib.nvbx_set_blocks(self._h, layer, _np_ptr(idx), idx.shape[0], _np_ptr(data)))

    def save_map(self, path):
        """Mapper::saveLayerCake (nvblox_node.cpp:1668): TSDF + colour + ESDF layers to one file."""
        self._check(self.lib.nvbx_save_map(self._h, str(path).encode()))

    def load_map(self, path):
        """Mapper::loadMap (nvblox_node.cpp:1703): replaces the map with the file's layers."""
        self._check(self.lib.nvbx_load_map(self._h, str(path).encode()))

    def synthetic_depth(self):
        r, c = C.c_int32(), C.c_int32()
        self._check(self.lib.nvbx_get_synthetic_depth(self._h, None, 0, C.byref(r), C.byref(c)))
        if r.value == 0:
            return None
        out = np.zeros((r.value, c.value), np.float32)
        self._check(self.lib.nvbx_get_synthetic_depth(self._h, _np_ptr(out), out.size, C.byref(r), C.byref(c)))
        return out

    def esdf_slice_image(self, unknown_value=1000.0):
        """EsdfSlicer::sliceLayerToDistanceImage for a host caller: ONE wait for the device (nvbx_esdf_slice_to_host sizes the image on the device);
        the buffer it is written into is kept and grows when the layer outgrows it."""
        r, c = C.c_int32(), C.c_int32()
        aabb = np.zeros(6, np.float32)
        buf = getattr(self, "_slice_buf", None)
        if buf is None:
            buf = self._slice_buf = np.empty(256 * 256, np.float32)
        rc = self.lib.nvbx_esdf_slice_to_host(self._h, unknown_value, _np_ptr(buf), buf.size, C.byref(r), C.byref(c), _np_ptr(aabb))
        if rc == -3:          # NVBX_E_CAPACITY: rows / cols are reported
            buf = self._slice_buf = np.empty(int(r.value * c.value * 1.5), np.float32)
            rc = self.lib.nvbx_esdf_slice_to_host(self._h, unknown_value, _np_ptr(buf), buf.size, C.byref(r), C.byref(c), _np_ptr(aabb))
        self._check(rc)
        return buf[:r.value * c.value].reshape(r.value, c.value).copy(), aabb

    def esdf_slice_image_device(self, unknown_value=1000.0):
        torch = self._torch
        r, c = C.c_int32(), C.c_int32()
        aabb = np.zeros(6, np.float32)
        self._check(self.lib.nvbx_esdf_slice_size(self._h, C.byref(r), C.byref(c), _np_ptr(aabb)))
        img = torch.empty((r.value, c.value), dtype=torch.float32, device="cuda:%d" % self.device)
        if img.numel():
            self._check(self.lib.nvbx_esdf_slice_to_image(self._h, unknown_value, C.c_void_p(img.data_ptr()), img.numel(), C.byref(r), C.byref(c), _np_ptr(aabb)))
        return img, aabb

    def occupancy_grid_from_slice(self, img_dev, unknown_value=1000.0):
        torch = self._torch
        grid = torch.empty(img_dev.shape, dtype=torch.int8, device=img_dev.device)
        self._check(self.lib.nvbx_occupancy_grid_from_slice(self._h, C.c_void_p(img_dev.data_ptr()), img_dev.shape[0], img_dev.shape[1], unknown_value, C.c_void_p(grid.data_ptr())))
        self.synchronize()
        return grid

    def pointcloud_from_slice(self, img_dev, aabb, slice_height, unknown_value=1000.0):
        torch = self._torch
        pts = torch.empty((img_dev.numel(), 4), dtype=torch.float32, device=img_dev.device)
        n = C.c_int32()
        a = np.ascontiguousarray(aabb, np.float32)
        self._check(self.lib.nvbx_pointcloud_from_slice(self._h, C.c_void_p(img_dev.data_ptr()), img_dev.shape[0], img_dev.shape[1], _np_ptr(a), float(slice_height), unknown_value, C.c_void_p(pts.data_ptr()), C.byref(n)))
        return pts[:n.value]

    def esdf_slice_image_combined(self, other, unknown_value=1000.0):
        """EsdfSlicer::sliceLayersToCombinedDistanceImage (nvblox_node.cpp:836-840) of this mapper and `other`; host image."""
        import torch
        r, c = C.c_int32(), C.c_int32(); aabb = (C.c_float * 6)()
        self._check(self.lib.nvbx_esdf_slice_combined_size(self._h, other._h, C.byref(r), C.byref(c), aabb))
        if r.value == 0:
            return np.zeros((0, 0), np.float32), np.array(list(aabb), np.float32)
        img = torch.empty((r.value, c.value), dtype=torch.float32, device="cuda:%d" % self.device)
        self._check(self.lib.nvbx_esdf_slice_combined_to_image(self._h, other._h, unknown_value, C.c_void_p(img.data_ptr()), img.numel(),
                                                               C.byref(r), C.byref(c), aabb))
        self.synchronize()
        return img.cpu().numpy(), np.array(list(aabb), np.float32)

    def backproject_depth(self, depth, cam, max_distance_m=0.0, T_L_C=None):
        """DepthImageBackProjector::backProjectOnGPU (+ transformPointcloudOnGPU if T_L_C is given): [n, 3] float32 host array."""
        import torch
        d, rows, cols = self._dev(depth, self._torch.float32), depth.shape[0], depth.shape[1]
        pts = torch.empty((rows * cols, 3), dtype=torch.float32, device=d.device)
        n = C.c_int64()
        self._check(self.lib.nvbx_backproject_depth(self._h, C.c_void_p(d.data_ptr()), rows, cols, C.byref(self._cam(cam)), max_distance_m,
                                                    C.c_void_p(pts.data_ptr()), rows * cols, C.byref(n)))
        pts = pts[:n.value]
        if T_L_C is not None and n.value:
            T = self._T(T_L_C)
            self._check(self.lib.nvbx_transform_pointcloud(self._h, _np_ptr(T), C.c_void_p(pts.data_ptr()), n.value, C.c_void_p(pts.data_ptr())))
            self.synchronize()
        return pts.cpu().numpy()

    def split_depth_by_mask(self, depth, mask, T_CM_CD, depth_cam, mask_cam, occlusion_threshold_m=0.25, overlay=False):
        """ImageMasker::splitImageOnGPU (MultiMapper::integrateDepth with a mask): device tensors (unmasked, masked[, overlay])."""
        torch = self._torch
        d = self._dev(depth, torch.float32); mk = self._dev(mask, torch.uint8)
        un = torch.empty_like(d); ma = torch.empty_like(d)
        ov = torch.empty(d.shape + (3,), dtype=torch.uint8, device=d.device) if overlay else None
        self._check(self.lib.nvbx_split_depth_by_mask(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], C.c_void_p(mk.data_ptr()),
                                                      mk.shape[0], mk.shape[1], _np_ptr(self._T(T_CM_CD)), C.byref(self._cam(depth_cam)),
                                                      C.byref(self._cam(mask_cam)), float(occlusion_threshold_m), C.c_void_p(un.data_ptr()),
                                                      C.c_void_p(ma.data_ptr()), C.c_void_p(ov.data_ptr()) if overlay else None))
        self.synchronize()       # (the outputs are torch tensors: torch's stream is not the mapper's)
        return (un, ma, ov) if overlay else (un, ma)

    def split_color_by_mask(self, rgb, mask):
        torch = self._torch
        c = self._dev(rgb, torch.uint8); mk = self._dev(mask, torch.uint8)
        un = torch.empty_like(c); ma = torch.empty_like(c)
        self._check(self.lib.nvbx_split_color_by_mask(self._h, C.c_void_p(c.data_ptr()), c.shape[0], c.shape[1], C.c_void_p(mk.data_ptr()),
                                                      C.c_void_p(un.data_ptr()), C.c_void_p(ma.data_ptr())))
        self.synchronize()
        return un, ma

    def device_view(self):
        """nvbx_device_view for the caller's own kernels (include/nvblox_hip_device.h)."""
        from ._lib import DeviceView
        v = DeviceView()
        self._check(self.lib.nvbx_get_device_view(self._h, C.byref(v)))
        return v

    def esdf_dense_grid(self, min_vox, size_vox, default_value):
        torch = self._torch
        mn = np.asarray(min_vox, np.int32); sz = np.asarray(size_vox, np.int32)
        out = torch.empty(tuple(int(s) for s in sz), dtype=torch.float32, device="cuda:%d" % self.device)
        self._check(self.lib.nvbx_esdf_dense_grid(self._h, _np_ptr(mn), _np_ptr(sz), default_value, C.c_void_p(out.data_ptr())))
        self.synchronize()
        return out.cpu().numpy()

    def mesh(self):
        """Mesh of the last update_color_mesh: dict block index tuple -> dict(vertices, normals, colors, triangles)."""
        nb, nv, nt = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.nvbx_mesh_sizes(self._h, C.byref(nb), C.byref(nv), C.byref(nt)))
        nb, nv, nt = nb.value, nv.value, nt.value
        idx = np.zeros((max(nb, 1), 3), np.int32); vo = np.zeros(nb + 1, np.int32); to = np.zeros(nb + 1, np.int32)
        v = np.zeros((max(nv, 1), 3), np.float32); n = np.zeros((max(nv, 1), 3), np.float32)
        c = np.zeros((max(nv, 1), 4), np.uint8); t = np.zeros((max(nt, 1), 3), np.int32)
        self._check(self.lib.nvbx_mesh_copy(self._h, _np_ptr(idx), _np_ptr(vo), _np_ptr(to), _np_ptr(v), _np_ptr(n), _np_ptr(c), _np_ptr(t)))
        out = {}
        for i in range(nb):
            out[tuple(int(q) for q in idx[i])] = dict(vertices=v[vo[i]:vo[i + 1]], normals=n[vo[i]:vo[i + 1]],
                                                      colors=c[vo[i]:vo[i + 1]], triangles=t[to[i]:to[i + 1]])
        return out

    # -- multi-GPU, one fused map: measurement exchange (nvbx_measure_depth / nvbx_apply_measurements; dist.MeasurementFusion)
    MEAS_BLOCK_BYTES = 4112

    def measure_depth(self, depth, T_L_C, cam, out_blocks, out_count):
        """This rank's camera -> measurement records into caller-owned device tensors: out_blocks uint8 [capacity, 4112], out_count int32 [1]."""
        d = self._dev(depth, self._torch.float32)
        T = self._T(T_L_C); k = self._cam(cam)
        self._check(self.lib.nvbx_measure_depth(self._h, C.c_void_p(d.data_ptr()), d.shape[0], d.shape[1], _np_ptr(T), C.byref(k),
                                                C.c_void_p(out_blocks.data_ptr()), C.c_void_p(out_count.data_ptr()), int(out_blocks.shape[0])))
        self._hold("_keep", [d])

    def apply_measurements(self, gathered, counts, owner_mod=0, owner_rank=0):
        """gathered uint8 [world, stride, 4112], counts int32 [world] (device): every camera's measurements applied in rank order."""
        self._check(self.lib.nvbx_apply_measurements(self._h, C.c_void_p(gathered.data_ptr()), C.c_void_p(counts.data_ptr()), int(gathered.shape[0]),
                                                     int(gathered.shape[1]), int(owner_mod), int(owner_rank)))

    # -- multi-GPU hooks
    def esdf_dirty_list(self, idx_out, count_out):
        """Write the Index3D list [cap,3] int32 + count [1] int32 of TSDF blocks dirtied since the last updateEsdf
        into caller-owned device tensors (async)."""
        self._check(self.lib.nvbx_esdf_dirty_list(self._h, C.c_void_p(idx_out.data_ptr()), C.c_void_p(count_out.data_ptr()), int(idx_out.shape[0])))

    def set_view_export(self, packed_tensor):
        """Register (or, with None, unregister) an int32 device tensor [1 + cap, 3]: every following integrate_depth writes the indices
        of the blocks it updates into it (row 0 = count) from inside its own launch.  The tensor must stay alive while registered."""
        if packed_tensor is None:
            self._check(self.lib.nvbx_set_view_export(self._h, None, 0)); self._view_export = None
        else:
            self._view_export = packed_tensor
            self._check(self.lib.nvbx_set_view_export(self._h, C.c_void_p(packed_tensor.data_ptr()), int(packed_tensor.shape[0]) - 1))

    def mark_esdf_dirty(self, idx_tensor, count_tensor, max_count):
        self._check(self.lib.nvbx_mark_esdf_dirty(self._h, C.c_void_p(idx_tensor.data_ptr()), C.c_void_p(count_tensor.data_ptr()), int(max_count)))

    def mark_esdf_dirty_gathered(self, gathered, world, self_rank, max_count, deferred=False):
        """gathered: int32 device tensor [world, 1 + max_count, 3] (row 0 = count); one launch for all peers -- or, deferred,
        no launch at all: the next integrate_color carries it (the tensor must stay alive and unchanged until then)."""
        fn = self.lib.nvbx_mark_esdf_dirty_gathered_deferred if deferred else self.lib.nvbx_mark_esdf_dirty_gathered
        self._check(fn(self._h, C.c_void_p(gathered.data_ptr()), int(world), int(self_rank), int(max_count)))

    # -- instrumentation
    def set_profiling(self, enable):
        self._check(self.lib.nvbx_set_profiling(self._h, int(enable)))

    def profile(self):
        import json
        buf = C.create_string_buffer(1 << 16)
        self._check(self.lib.nvbx_get_profile(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())
