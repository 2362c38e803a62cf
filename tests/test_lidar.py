"""Spinning-LiDAR projective integration (BASELINE.json configs[4]): sensor-model known answers on CPU, HIP vs oracle on GPU."""
import math

import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

SMALL_LIDAR = (256, 16, 0.1, -math.radians(15.0), math.radians(15.0))


def test_lidar_model_known_answers(oracle_mod):
    """Beam (k, j) projects to the pixel centre (j + 0.5, k + 0.5) -- what checkLidarPointcloud asserts of a consistent
    model (conversions/pointcloud_conversions.cu:73-97) -- with elevation = asin(z / r), azimuth = atan2(y, x)
    (scripts/calculate_lidar_params.py:50-58)."""
    for lidar in (S.SPINNING_LIDAR, SMALL_LIDAR, (2048, 64, 0.01, -0.285, 0.298)):      # last: nvblox_os1.yaml:7-12
        dirs = S.lidar_beam_dirs(lidar)
        rows, cols = dirs.shape[:2]
        for k, j in [(0, 0), (rows - 1, cols - 1), (rows // 2, cols // 2), (3, 7), (rows - 2, 1)]:
            uv = oracle_mod.lidar_project(lidar, 17.0 * dirs[k, j])
            assert uv is not None
            assert abs(uv[0] - (j + 0.5)) < 2e-3 and abs(uv[1] - (k + 0.5)) < 2e-3
        # outside the vertical field of view / below the minimum range
        up = np.array([0.1, 0.0, 1.0]); assert oracle_mod.lidar_project(lidar, up) is None
        assert oracle_mod.lidar_project(lidar, 0.5 * lidar[2] * dirs[1, 1]) is None


def test_shared_atan2_accuracy(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(3)
    worst = 0.0
    for y, x in rng.normal(size=(5000, 2)).astype(np.float32):
        worst = max(worst, abs(L.orc_atan2f(float(y), float(x)) - math.atan2(float(y), float(x))))
    for y, x in [(0.0, 1.0), (0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (0.0, 0.0)]:
        worst = max(worst, abs(L.orc_atan2f(y, x) - math.atan2(y, x)))
    assert worst < 5e-7


def test_depth_image_from_pointcloud_oracle(oracle_mod):
    """Point cloud -> range image (depthImageFromPointcloudKernel, pointcloud_conversions.cu:118-150): one point per beam
    reproduces the range image; NaN points are skipped."""
    sc = S.LidarScene(n_boxes=10, extent=40.0)
    T = S.lidar_pose(0)
    img = S.render_lidar(sc, T, SMALL_LIDAR, max_range=60.0)
    dirs = S.lidar_beam_dirs(SMALL_LIDAR)
    pts = (dirs * img[..., None]).reshape(-1, 3).astype(np.float32)
    pts = pts[img.reshape(-1) > 0]
    pts = np.concatenate([pts, np.full((3, 3), np.nan, np.float32)])
    back = oracle_mod.depth_image_from_pointcloud(pts, SMALL_LIDAR)
    assert back.shape == img.shape
    assert np.abs(back - img).max() < 1e-4 and ((back > 0) == (img > 0)).all()


def test_lidar_tsdf_against_analytic_ground(oracle_mod):
    """Independent check of the oracle's LiDAR TSDF: near the ground plane z = 0, at steep incidence (where the nearest-beam
    acceptance radius of half a voxel bounds the range error by ~0.09 m), the projective distance must agree with the
    true distance to the plane along the voxel's own beam."""
    po = oracle_mod.default_params(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=2)
    o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=0)
    T = S.lidar_pose(0, height=2.0)
    wide = (256, 32, 0.1, -math.radians(60.0), math.radians(10.0))
    img = S.render_lidar(sc, T, wide, max_range=30.0)
    n = o.integrate_lidar_depth(img, T, wide)
    assert n > 50
    checked = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        if idx[2] != 0:
            continue
        b = o.get_block(oracle_mod.L_TSDF, idx).reshape(8, 8, 8)      # [x][y][z]
        w = b["weight"]; d = b["distance"]
        for x, y, z in zip(*np.nonzero(w > 0)):
            pz = (idx[2] * 8 + z + 0.5) * 0.1
            px = (idx[0] * 8 + x + 0.5) * 0.1 - float(T[0, 3]); py = (idx[1] * 8 + y + 0.5) * 0.1 - float(T[1, 3])
            dz = pz - float(T[2, 3])
            r = math.sqrt(px * px + py * py + dz * dz)
            if abs(d[x, y, z]) < 0.35 and -dz / r > 0.5:
                true_along_beam = pz * r / (-dz)          # distance from the voxel to the plane along its own beam
                assert abs(d[x, y, z] - true_along_beam) < 0.12, (idx, x, y, z, d[x, y, z], true_along_beam)
                checked += 1
    assert checked > 100


@pytest.mark.gpu
def test_lidar_parity_small(oracle_mod, hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=40.0, raycast_subsampling_factor=2)
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 16); o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    for i in range(3):
        T = S.lidar_pose(i * 7)
        img = S.render_lidar(sc, T, SMALL_LIDAR, max_range=60.0)
        g.integrate_lidar_depth(img, T, SMALL_LIDAR); o.integrate_lidar_depth(img, T, SMALL_LIDAR)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    ig = g.block_indices(M.LAYER_TSDF); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io) and len(io) > 1000
    bg, found = g.get_blocks(M.LAYER_TSDF, ig)
    assert found.all()
    for k, idx in enumerate(io):
        bo = o.get_block(oracle_mod.L_TSDF, idx)
        assert np.abs(bg[k]["distance"] - bo["distance"]).max() <= 1e-4, idx
        assert np.abs(bg[k]["weight"] - bo["weight"]).max() <= 1e-4, idx
    assert g.counters()["capacity_overflow"] == 0
    # ESDF + mesh run on a LiDAR-built map like on a camera-built one
    g.update_esdf(); o.update_esdf()
    sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("lidar,tilt", [((256, 48, 0.1, -math.radians(55.0), math.radians(50.0)), (0.0, 0.0)),          # wide vertical FOV: |sin el| > 0.5
                                        ((256, 48, 0.1, -math.radians(55.0), math.radians(50.0)), (11.0, -17.0)),      # ... on a rolled / pitched sensor
                                        (SMALL_LIDAR, (23.0, 9.0))])
def test_lidar_parity_wide_fov_and_tilted_sensor(oracle_mod, hip_lib, lidar, tilt):
    """The elevation's polynomial covers |sin el| <= 0.5 and falls back to atan2 beyond (csrc/nvbx_lidar_math.h); the integrators add a
    rotated voxel offset to a per-block origin (nvbx_internal.h sensor_block_origin).  Both on inputs the upright 15-degree sensor of the
    other tests never produces: beams up to 55 degrees, a sensor frame rotated about all three axes."""
    from isaac_ros_nvblox_amd import mapper as M
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=25.0, raycast_subsampling_factor=2)
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 16); o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    roll, pitch = math.radians(tilt[0]), math.radians(tilt[1])
    Rx = np.array([[1, 0, 0], [0, math.cos(roll), -math.sin(roll)], [0, math.sin(roll), math.cos(roll)]])
    Ry = np.array([[math.cos(pitch), 0, math.sin(pitch)], [0, 1, 0], [-math.sin(pitch), 0, math.cos(pitch)]])
    n_steep = 0
    for i in range(3):
        T = S.lidar_pose(i * 7).astype(np.float64)
        T[:3, :3] = T[:3, :3] @ Ry @ Rx
        T = T.astype(np.float32)
        img = S.render_lidar(sc, T, lidar, max_range=40.0)
        n_steep += int((img[:6] > 0).sum() + (img[-6:] > 0).sum())
        g.integrate_lidar_depth(img, T, lidar); o.integrate_lidar_depth(img, T, lidar)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    assert n_steep > 100                                                                  # the steepest beams do hit something
    ig = g.block_indices(M.LAYER_TSDF); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io) and len(io) > 500
    bg, found = g.get_blocks(M.LAYER_TSDF, ig)
    assert found.all()
    nobs = 0
    for k, idx in enumerate(io):
        b = o.get_block(oracle_mod.L_TSDF, idx)
        assert np.array_equal(bg[k]["distance"], b["distance"]) and np.array_equal(bg[k]["weight"], b["weight"]), idx     # bit for bit
        nobs += int((b["weight"] > 0).sum())
    assert nobs > 50000


@pytest.mark.gpu
@pytest.mark.parametrize("thr", [(2.0, 0.5), (0.3, 1.5)])
def test_lidar_parity_noisy_ranges_with_dropouts(oracle_mod, hip_lib, thr):
    """Range noise and 3 % missing returns: four-beam neighbourhoods that disagree or have holes fall from the bilinear blend to the
    nearest-beam rule (LidarSensor::sample, tsdf.hip) for a large share of the voxels -- the branch the clean scenes of the other tests
    rarely take.  Bit for bit against the oracle, with the thresholds in two positions."""
    from isaac_ros_nvblox_amd import mapper as M
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=2,
              lidar_linear_interpolation_max_allowable_difference_vox=thr[0], lidar_nearest_interpolation_max_allowable_dist_to_ray_vox=thr[1])
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 16); o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    rng = np.random.default_rng(11)
    for i in range(3):
        T = S.lidar_pose(i * 7)
        img = S.render_lidar(sc, T, SMALL_LIDAR, max_range=40.0)
        img = np.where(img > 0, img + rng.normal(0.0, 0.03, img.shape).astype(np.float32), 0.0).astype(np.float32)
        img[rng.random(img.shape) < 0.03] = 0.0
        if i == 1:                                     # a driver hiccup: a few non-finite / negative ranges
            bad = rng.integers(0, img.size, 60)
            img.reshape(-1)[bad[:20]] = np.nan; img.reshape(-1)[bad[20:40]] = np.inf; img.reshape(-1)[bad[40:]] = -2.0
        g.integrate_lidar_depth(img, T, SMALL_LIDAR); o.integrate_lidar_depth(img, T, SMALL_LIDAR)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    ig = g.block_indices(M.LAYER_TSDF); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io) and len(io) > 500
    bg, found = g.get_blocks(M.LAYER_TSDF, ig)
    assert found.all()
    nobs = 0
    for k, idx in enumerate(io):
        b = o.get_block(oracle_mod.L_TSDF, idx)
        assert np.array_equal(bg[k]["distance"], b["distance"]) and np.array_equal(bg[k]["weight"], b["weight"]), idx
        nobs += int((b["weight"] > 0).sum())
        assert np.isfinite(b["distance"]).all() and np.isfinite(b["weight"]).all()
    assert nobs > 20000


@pytest.mark.gpu
def test_lidar_full_config_properties(hip_lib):
    """BASELINE.json configs[4] shape: 1024 x 64 beams, 0.10 m voxels, 200 m range (too slow for the scalar oracle at full
    size, so size-independent properties): integrating the same scan twice leaves the block set unchanged and doubles
    the weights up to the clamp; the block set of a scan is a superset of the blocks containing its surface points."""
    from isaac_ros_nvblox_amd import mapper as M
    pg = M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2, weighting_mode=0)
    g = M.Mapper(pg, block_capacity=1 << 19)
    sc = S.LidarScene()
    T = S.lidar_pose(0)
    img = S.render_lidar(sc, T, S.SPINNING_LIDAR, max_range=200.0)
    g.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
    c1 = g.counters(); idx1 = g.block_indices(M.LAYER_TSDF)
    assert c1["capacity_overflow"] == 0 and c1["tsdf_blocks_in_view"] == len(idx1) and len(idx1) > 20000
    g.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
    idx2 = g.block_indices(M.LAYER_TSDF)
    assert np.array_equal(idx1, idx2)                                  # idempotent block set
    sample = idx1[:: max(1, len(idx1) // 300)]
    b, found = g.get_blocks(M.LAYER_TSDF, sample)
    assert found.all()
    w = b["weight"]
    assert set(np.unique(w).tolist()) <= {0.0, 2.0}                     # constant weight 1, two identical scans
    # surface points of the sub-sampled rays lie in allocated blocks
    dirs = S.lidar_beam_dirs(S.SPINNING_LIDAR)
    Td = np.asarray(T, np.float64)
    pts = (dirs * img[..., None])[::2, ::2].reshape(-1, 3)
    pts = pts[(img[::2, ::2].reshape(-1) > 0)]
    pw = pts @ Td[:3, :3].T + Td[:3, 3]
    bi = np.floor(pw / 0.8).astype(np.int64)
    have = H.idx_set(idx1)
    miss = [tuple(q) for q in bi[::17].tolist() if tuple(q) not in have]
    assert len(miss) == 0, miss[:5]


@pytest.mark.gpu
def test_lidar_config_with_automatic_and_growing_pools(hip_lib):
    """configs[4] through the capacity the facade's reference-signature MultiMapper constructor uses (0 = automatic) and through pools
    that start far too small for the map and grow: no block is dropped once the pools have caught up, the maps agree."""
    from isaac_ros_nvblox_amd import mapper as M
    pg = M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2, weighting_mode=0)
    sc = S.LidarScene()
    ga = M.Mapper(pg, block_capacity=0)                       # automatic
    gg = M.Mapper(pg, block_capacity=1 << 19)
    assert ga.capacity >= (1 << 16)
    for i in range(3):
        T = S.lidar_pose(i * 3)
        img = S.render_lidar(sc, T, S.SPINNING_LIDAR, max_range=200.0)
        ga.integrate_lidar_depth(img, T, S.SPINNING_LIDAR); gg.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
        ga.synchronize()
    assert ga.counters()["capacity_overflow"] == 0 and gg.counters()["capacity_overflow"] == 0
    ia, ig = ga.block_indices(M.LAYER_TSDF), gg.block_indices(M.LAYER_TSDF)
    assert np.array_equal(ia, ig) and len(ia) > 100000
    sample = ia[:: max(1, len(ia) // 400)]
    ba, _ = ga.get_blocks(M.LAYER_TSDF, sample); bg, _ = gg.get_blocks(M.LAYER_TSDF, sample)
    assert np.array_equal(ba["distance"], bg["distance"]) and np.array_equal(ba["weight"], bg["weight"])


def _moving_scan(n=4000, seed=3):
    """A sensor translating 0.3 m and yawing 6 degrees during a 100 ms scan of a static point set."""
    rng = np.random.default_rng(seed)
    world = rng.uniform([-8, -8, -1], [8, 8, 3], (n, 3))
    t_ms = np.sort(rng.uniform(0.0, 100.0, n)).astype(np.float32)
    def pose(a):
        yaw = np.deg2rad(6.0) * a
        T = np.eye(4); c, s = np.cos(yaw), np.sin(yaw)
        T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; T[:3, 3] = [0.3 * a, 0.05 * a, 0.0]
        return T
    pts_meas = np.stack([(np.linalg.inv(pose(t / 100.0)) @ np.append(w, 1.0))[:3] for w, t in zip(world, t_ms)]).astype(np.float32)
    truth = world.astype(np.float32)                          # the sensor frame at scan start is the world frame here
    return pts_meas, t_ms, pose(0.0).astype(np.float32), pose(1.0).astype(np.float32), truth


def test_motion_compensation_oracle_against_exact_motion(oracle_mod):
    """[U] LiDAR motion compensation: de-skewing with the interpolated pose recovers the static scene to within a millimetre for a
    realistic scan motion (the normalised-quaternion interpolation deviates from the exact screw motion by far less)."""
    pts, t_ms, T0, T1, truth = _moving_scan()
    out = oracle_mod.motion_compensate_pointcloud(pts, t_ms, T0, T1, 100.0)
    raw_err = np.linalg.norm(pts - truth, axis=1)
    err = np.linalg.norm(out - truth, axis=1)
    assert raw_err.max() > 0.3 and err.max() < 1e-3
    # identity motion is the identity; times outside the scan clamp
    same = oracle_mod.motion_compensate_pointcloud(pts, t_ms, T0, T0, 100.0)
    assert np.abs(same - pts).max() < 1e-6
    clamp = oracle_mod.motion_compensate_pointcloud(pts[:2], np.array([-5.0, 500.0], np.float32), T0, T1, 100.0)
    assert np.allclose(clamp[0], pts[0], atol=1e-6) and np.allclose(clamp[1], oracle_mod.motion_compensate_pointcloud(pts[1:2], np.array([100.0], np.float32), T0, T1, 100.0)[0])


def _random_motions(rng, n):
    from scipy.spatial.transform import Rotation
    out = []
    for _ in range(n):
        T0 = np.eye(4); T0[:3, :3] = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(); T0[:3, 3] = rng.uniform(-20, 20, 3)
        d = np.eye(4); d[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * np.deg2rad(rng.uniform(0.0, 12.0)) / np.sqrt(3.0)).as_matrix(); d[:3, 3] = rng.uniform(-0.6, 0.6, 3)
        out.append((T0.astype(np.float32), (T0 @ d).astype(np.float32)))
    return out


def test_motion_compensation_against_an_independent_float64_model(oracle_mod):
    """The per-point arithmetic of the motion compensation is SHARED by the kernel and the checker (csrc/nvbx_motion_math.h: that is what makes them
    bit-identical) -- so a slip in it would be invisible to every HIP-vs-checker comparison.  tests/motion_independent.py
    restates the model in float64 on numpy + scipy.Rotation; the checker must agree with it to float32 accuracy on random poses anywhere in
    the map, relative rotations up to 12 degrees, points out to 150 m, times inside and outside the scan."""
    import motion_independent as MI
    rng = np.random.default_rng(11)
    worst = 0.0
    for T0, T1 in _random_motions(rng, 12):
        pts = (rng.normal(size=(20000, 3)) * rng.uniform(1.0, 50.0)).astype(np.float32)
        t_ms = rng.uniform(-10.0, 110.0, len(pts)).astype(np.float32)
        got = oracle_mod.motion_compensate_pointcloud(pts, t_ms, T0, T1, 100.0)
        want = MI.compensate(pts, t_ms, T0, T1, 100.0)
        err = np.linalg.norm(got - want, axis=1) / np.maximum(1.0, np.linalg.norm(pts.astype(np.float64), axis=1))
        worst = max(worst, float(err.max()))
    assert worst < 5e-6, worst            # a few float32 ulps of the point's magnitude


@pytest.mark.gpu
def test_motion_compensation_gpu_against_the_independent_model(hip_lib):
    import motion_independent as MI
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=256)
    rng = np.random.default_rng(12)
    for T0, T1 in _random_motions(rng, 6):
        pts = (rng.normal(size=(50000, 3)) * rng.uniform(1.0, 50.0)).astype(np.float32)
        t_ms = rng.uniform(-10.0, 110.0, len(pts)).astype(np.float32)
        got = g.motion_compensate_pointcloud(pts, t_ms, T0, T1, 100.0).cpu().numpy()
        want = MI.compensate(pts, t_ms, T0, T1, 100.0)
        err = np.linalg.norm(got - want, axis=1) / np.maximum(1.0, np.linalg.norm(pts.astype(np.float64), axis=1))
        assert err.max() < 5e-6, err.max()


@pytest.mark.gpu
def test_motion_compensation_gpu_parity(oracle_mod, hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=256)
    pts, t_ms, T0, T1, truth = _moving_scan(20000, seed=4)
    out = g.motion_compensate_pointcloud(pts, t_ms, T0, T1, 100.0).cpu().numpy()
    want = oracle_mod.motion_compensate_pointcloud(pts, t_ms, T0, T1, 100.0)
    assert np.array_equal(out, want)                          # shared arithmetic: bit-exact
    assert np.linalg.norm(out - truth, axis=1).max() < 1e-3


@pytest.mark.gpu
def test_lidar_depth_then_camera_colour_parity(oracle_mod, hip_lib):
    """A LiDAR scan builds the TSDF, a camera frame colours it (use_lidar + a colour camera): the LiDAR integration leaves the
    colour integrator's per-block band flags stale and the colour launch repairs them from the TSDF -- same colour layer as the
    oracle; a second colour frame (flags repaired) and a camera depth frame in between (flags kept exact) as well."""
    from isaac_ros_nvblox_amd import mapper as M
    from test_gpu_parity import make_pair, compare_layer
    lidar = (256, 32, 0.1, -np.deg2rad(30.0), np.deg2rad(30.0))
    Mm, g, o = make_pair(oracle_mod, lidar_max_integration_distance_m=8.0, raycast_subsampling_factor=2)
    sc = S.Scene()
    T_l = np.eye(4, dtype=np.float32); T_l[:3, 3] = [0.2, -0.1, 1.3]
    rng_img = S.render_lidar(sc, T_l, lidar, max_range=8.0)
    g.integrate_lidar_depth(rng_img, T_l, lidar); o.integrate_lidar_depth(rng_img, T_l, lidar)
    cam = H.SMALL_CAM
    frames = H.frames(3, cam, color=True, stride=15)
    d, rgb, T = frames[0]
    g.integrate_color(rgb, T, cam); o.integrate_color(rgb, T, cam)
    n, _ = compare_layer(Mm, g, o, Mm.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    assert n > 20
    g.integrate_color(frames[1][1], frames[1][2], cam); o.integrate_color(frames[1][1], frames[1][2], cam)
    g.integrate_depth(frames[2][0], frames[2][2], cam); o.integrate_depth(frames[2][0], frames[2][2], cam)
    g.integrate_lidar_depth(rng_img, T_l, lidar); o.integrate_lidar_depth(rng_img, T_l, lidar)
    g.integrate_color(frames[2][1], frames[2][2], cam); o.integrate_color(frames[2][1], frames[2][2], cam)
    compare_layer(Mm, g, o, Mm.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    compare_layer(Mm, g, o, Mm.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))


@pytest.mark.gpu
def test_depth_image_from_pointcloud_gpu_parity_with_degenerate_points(oracle_mod, hip_lib):
    """depthImageFromPointcloud (pointcloud_conversions.cu:118-150) on the GPU == the oracle, for a scan plus the points a driver can
    hand over by accident: NaNs, the origin, points a nanometre / 10^20 m away, points exactly on the sensor's axes.  Distinct pixels per
    point (last-writer-wins ties are order-dependent), so the two images must be identical."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 10)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    T = S.lidar_pose(3)
    img = S.render_lidar(sc, T, SMALL_LIDAR, max_range=60.0)
    dirs = S.lidar_beam_dirs(SMALL_LIDAR)
    pts = (dirs * img[..., None]).reshape(-1, 3)[img.reshape(-1) > 0].astype(np.float32)
    odd = np.float32([[np.nan, 1, 1], [0, 0, 0], [1e-12, 0, 0], [0, 0, 5.0], [0, 0, -5.0], [1e20, 1e20, 0], [3e-39, 0, 2.0], [np.inf, 0, 0], [0.05, 0.0, 0.0]])          # (all of them are outside the model: the image is the scan's)
    allp = np.concatenate([pts, odd]).astype(np.float32)
    got = g.depth_image_from_pointcloud(torch.from_numpy(allp).cuda(), SMALL_LIDAR).cpu().numpy()
    want = oracle_mod.depth_image_from_pointcloud(allp, SMALL_LIDAR)
    assert got.shape == want.shape == img.shape
    assert np.array_equal(got, want)
    assert (got > 0).sum() >= (img > 0).sum() * 0.98 and np.isfinite(got).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_azimuth_sector_split_reproduces_the_single_sensor_map_away_from_the_cuts(hip_lib, world):
    """bench.py --workload lidar --gpus N (BASELINE.json configs[4] "1 and 8 GPUs"): rank r integrates only the beams of azimuth sector r.
    Every voxel whose four interpolation taps lie inside one sector (two or more columns away from both cuts) must come out of that
    rank's map exactly as out of the single-sensor map; together the sector maps observe everything the full scan observes except in the
    columns at the cuts, and the sector views add up to the full view (blocks on a cut are shared)."""
    from isaac_ros_nvblox_amd import mapper as M
    lidar = SMALL_LIDAR
    cols, rows = lidar[0], lidar[1]
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=1)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    T = S.lidar_pose(5)
    img = S.render_lidar(sc, T, lidar, max_range=40.0)
    full = M.Mapper(M.default_params(**kw), block_capacity=1 << 16)
    full.integrate_lidar_depth(img, T, lidar)
    fidx = full.block_indices(M.LAYER_TSDF); fb, _ = full.get_blocks(M.LAYER_TSDF, fidx)
    fmap = {tuple(i): fb[k] for k, i in enumerate(fidx.tolist())}
    # azimuth column of every voxel centre of the full map (numpy, float64: only used to pick voxels far from the cuts)
    Tinv = np.linalg.inv(np.asarray(T, np.float64))
    vs = 0.1
    g3 = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)          # x, y, z ; linear = z + 8y + 64x
    lin = g3[:, 2] + 8 * g3[:, 1] + 64 * g3[:, 0]
    covered = 0; checked = 0; views = []
    sector_maps = []
    for r in range(world):
        lo, hi = r * cols // world, (r + 1) * cols // world
        part = np.where((np.arange(cols) >= lo) & (np.arange(cols) < hi), img, np.float32(0.0)).astype(np.float32)
        g = M.Mapper(M.default_params(**kw), block_capacity=1 << 16)
        g.integrate_lidar_depth(part, T, lidar)
        views.append(H.idx_set(g.last_view()))
        idx = g.block_indices(M.LAYER_TSDF); b, _ = g.get_blocks(M.LAYER_TSDF, idx)
        sector_maps.append(({tuple(i): b[k] for k, i in enumerate(idx.tolist())}, lo, hi))
    assert set().union(*views) == H.idx_set(full.last_view())
    for bi, blk in fmap.items():
        centres = (np.asarray(bi, np.float64)[None, :] * 8 + g3 + 0.5) * vs
        pc = centres @ Tinv[:3, :3].T + Tinv[:3, 3]
        u = (np.arctan2(pc[:, 1], pc[:, 0]) + np.pi) / (2 * np.pi / cols) + 0.5
        u = np.where(u >= cols, u - cols, u)
        w_full = blk["weight"][lin]; d_full = blk["distance"][lin]
        for smap, lo, hi in sector_maps:
            inside = (u > lo + 2.0) & (u < hi - 2.0) & (w_full > 0)
            if not inside.any():
                continue
            assert bi in smap
            sb = smap[bi]
            assert np.array_equal(sb["weight"][lin][inside], w_full[inside]) and np.array_equal(sb["distance"][lin][inside], d_full[inside]), bi
            checked += int(inside.sum())
        covered += int((w_full > 0).sum())
    assert checked > 0.7 * covered and covered > 50000             # (most of the map is more than two columns from a cut)


@pytest.mark.gpu
def test_lidar_hip_map_against_the_independent_model(hip_lib):
    """The PRODUCT (not the oracle) against tests/lidar_independent.py -- numpy float64 with libm, no code shared with csrc/: one
    configs[4] scan into an empty map with constant weighting, then for > 3 x 10^5 voxels of blocks at all ranges the voxel must be
    updated exactly when the independent model yields a measurement with sdf >= -truncation, and hold clamp(measured - range) within
    2e-4 m; voxels whose decision hangs on the last bits (pixel-bin edges, thresholds) are left out by the model's own margins."""
    import lidar_independent as LI
    from isaac_ros_nvblox_amd import mapper as M
    vs = 0.1; trunc = 4.0 * vs
    pg = M.default_params(voxel_size=vs, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2, weighting_mode=0)
    g = M.Mapper(pg, block_capacity=1 << 18)
    sc = S.LidarScene(); T = S.lidar_pose(7)
    img = S.render_lidar(sc, T, S.SPINNING_LIDAR, max_range=200.0)
    g.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
    idx = g.block_indices(M.LAYER_TSDF)
    assert len(idx) > 100000
    rng = np.random.default_rng(2)
    sel = idx[np.sort(rng.choice(len(idx), 800, replace=False))]
    b, found = g.get_blocks(M.LAYER_TSDF, sel)
    assert found.all()
    vx, vy, vz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")          # voxel order z + 8y + 64x
    off = np.stack([vx.ravel(), vy.ravel(), vz.ravel()], 1)
    centres = ((sel[:, None, :].astype(np.float64) * 8 + off[None, :, :]) + 0.5) * vs            # [blocks, 512, 3], world frame
    Td = np.asarray(T, np.float64)
    ps = (centres.reshape(-1, 3) - Td[:3, 3]) @ Td[:3, :3]                                       # sensor frame: R^T (c - t)
    ref = LI.sample(S.SPINNING_LIDAR, img, ps, 2.0 * vs, 0.5 * vs, 200.0)
    sdf = ref["ds"] - ref["r"]
    upd = (ref["branch"] > 0) & (sdf >= -trunc)
    robust = (ref["margin_px"] > 3e-3) & (ref["margin_m"] > 3e-4) & ((ref["branch"] == 0) | (np.abs(sdf + trunc) > 3e-4))
    w = b["weight"].reshape(-1); d = b["distance"].reshape(-1)
    assert robust.sum() > 300000 and robust.mean() > 0.85
    assert np.array_equal((w > 0)[robust], upd[robust]), int(((w > 0) != upd)[robust].sum())
    k = robust & upd
    assert set(np.unique(w[k]).tolist()) == {1.0}
    assert np.abs(d[k] - np.clip(sdf[k], -trunc, trunc)).max() < 2e-4
    far = np.hypot(ps[:, 0], ps[:, 1]) > 120.0
    assert (k & far).sum() > 1000 and (k & (ref["branch"] == 1)).sum() > 20000 and (k & (ref["branch"] == 2)).sum() > 5000
