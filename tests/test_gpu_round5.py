"""Round 5: the LiDAR view calculation over the dense grid against its fallbacks, and the host slice that takes one wait.

* `k_mark_view_grid` / `k_scan_view_grid` / `k_resolve_view` are the default and every LiDAR test runs through them; here the same
  parity tests run once more (a) with the hash path (`NVBX_LIDAR_VIEW_GRID=0`: `k_mark_view<Lidar>`) and (b) with a box of 12 blocks reach
  (`NVBX_VIEW_GRID_REACH=12`: most of a 25-40 m scan lies OUTSIDE the box and takes `mark_block`, block by block, beside the grid) -- the switches are
  read once per process, hence subprocesses;
* `nvbx_esdf_slice_to_host` (size + image written into pinned host memory by the slicing launch itself) against the two-step device slice, through
  growth of the layer, the capacity error and an empty layer."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [dict(NVBX_LIDAR_VIEW_GRID="0"), dict(NVBX_VIEW_GRID_REACH="12"), dict(NVBX_VIEW_GRID_MAX_MB="1")],
                         ids=["hash_path", "small_box_spills", "memory_cap_falls_back"])
def test_lidar_view_calculation_fallbacks_keep_parity(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_lidar.py"), "-m", "gpu", "-x", "-q", "-k",
                        "parity_small or wide_fov or noisy_ranges or depth_then_camera"], capture_output=True, text=True, timeout=1500, env=e, cwd=ROOT)
    assert p.returncode == 0 and " passed" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])


def test_view_grid_is_all_zero_between_scans_and_survives_moving_sensors(oracle_mod, hip_lib):
    """The grid is anchored at the sensor's block of each scan and must be all-zero when a scan's launches are done: scans from poses far apart (the box
    moves by tens of blocks), interleaved with camera frames and a clear(), keep giving the checker's views."""
    from isaac_ros_nvblox_amd import mapper as M
    lidar = (256, 16, 0.1, -0.26, 0.26)
    pg = M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=2, max_integration_distance_m=6.0)
    po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 16); o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    cam = H.SMALL_CAM
    fr = H.frames(2, cam, stride=17, color=False)
    for i, shift in enumerate([(0.0, 0.0), (17.3, -9.1), (-22.7, 4.4), (0.4, 0.3), (17.3, -9.1)]):
        T = S.lidar_pose(i * 5).copy(); T[0, 3] += shift[0]; T[1, 3] += shift[1]
        img = S.render_lidar(sc, T, lidar, max_range=45.0)
        g.integrate_lidar_depth(img, T, lidar); o.integrate_lidar_depth(img, T, lidar)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view()), i
        if i == 1:
            d, _, Tc = fr[0]; g.integrate_depth(d, Tc, cam); o.integrate_depth(d, Tc, cam)
            assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
        if i == 2:
            g.clear(); o = oracle_mod.OracleMap(po)
    assert np.array_equal(g.block_indices(M.LAYER_TSDF), o.block_indices(oracle_mod.L_TSDF))
    assert g.counters()["capacity_overflow"] == 0


def test_host_slice_in_one_wait_equals_the_two_step_device_slice(hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    g = M.Mapper(M.default_params(), block_capacity=1 << 13)
    lib = g.lib
    img0, aabb0 = g.esdf_slice_image()
    assert img0.shape == (0, 0)                                   # no ESDF block yet: rows = cols = 0, nothing copied
    for k, (d, rgb, T) in enumerate(H.frames(6, cam, stride=9)):
        g.integrate_depth(d, T, cam); g.integrate_color(rgb, T, cam); g.update_esdf()
        host, aabb_h = g.esdf_slice_image(unknown_value=777.0)    # held-back work is replayed first, then ONE launch + ONE wait
        dev, aabb_d = g.esdf_slice_image_device(unknown_value=777.0)
        assert host.shape == tuple(dev.shape) and host.size > 0
        assert np.array_equal(host, dev.cpu().numpy()) and np.array_equal(aabb_h, aabb_d), k
    # capacity: a buffer that is too small reports the size and copies nothing
    r, c = C.c_int32(), C.c_int32(); aabb = (C.c_float * 6)()
    small = np.full(16, -5.0, np.float32)
    rc = lib.nvbx_esdf_slice_to_host(g._h, C.c_float(777.0), small.ctypes.data_as(C.c_void_p), small.size, C.byref(r), C.byref(c), aabb)
    assert rc == -3 and (r.value, c.value) == host.shape and (small == -5.0).all()
    rc = lib.nvbx_esdf_slice_to_host(g._h, C.c_float(777.0), None, 0, C.byref(r), C.byref(c), aabb)
    assert rc == -3 and (r.value, c.value) == host.shape
