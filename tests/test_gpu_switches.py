"""GPU parity for every [U] open choice (SURVEY.md 8a "open choices the oracle must expose as switches").

The arithmetic of this path lives in the absent nvblox core, so each recollection that could be wrong is a switch on BOTH sides
(nvbx_mapper_params / OrcParams, same field names).  Every test below runs the HIP path against the oracle with the switch in each
position -- and checks that the positions really differ on the chosen input, so the switch is live.  Pinning to the real core is
then a flag flip, not a rewrite.  Reference anchors: mapper_initialization.cpp:31-42 (weighting modes), :246-380 (integrator knobs).
"""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_gpu_parity import TOL, compare_layer, make_pair

pytestmark = pytest.mark.gpu


def tsdf_map(g, M):
    idx = g.block_indices(M.LAYER_TSDF)
    b, found = g.get_blocks(M.LAYER_TSDF, idx)
    return {tuple(i): b[k].copy() for k, i in enumerate(idx.tolist())}


def maps_differ(a, b, field="weight"):
    if set(a) != set(b):
        return True
    return any(not np.array_equal(a[k][field], b[k][field]) or not np.array_equal(a[k]["distance"], b[k]["distance"]) for k in a)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("weighting_mode", [0, 1, 2, 3, 4, 5])
def test_all_six_weighting_modes_both_formula_sets(oracle_mod, hip_lib, weighting_mode, variant):
    """WeightingFunctionType x tsdf_weighting_variant (mapper_initialization.cpp:31-42): 12 combinations, HIP == oracle."""
    M, g, o = make_pair(oracle_mod, weighting_mode=weighting_mode, tsdf_weighting_variant=variant)
    for d, rgb, T in H.frames(5, H.SMALL_CAM, color=False, stride=7):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 100


def test_weighting_formula_sets_are_distinct(hip_lib):
    """The two formula sets differ for the four non-trivial modes and agree for constant / inverse-square."""
    from isaac_ros_nvblox_amd import mapper as M
    fr = H.frames(3, H.SMALL_CAM, color=False, stride=7)
    for mode in range(6):
        maps = []
        for variant in (0, 1):
            g = M.Mapper(M.default_params(weighting_mode=mode, tsdf_weighting_variant=variant), block_capacity=1 << 14)
            for d, rgb, T in fr:
                g.integrate_depth(d, T, H.SMALL_CAM)
            maps.append(tsdf_map(g, M))
        assert maps_differ(maps[0], maps[1]) == (mode in (1, 3, 4, 5)), mode


def _edge_case_frame():
    """A fronto-parallel plane seen by an identity-pose camera on a binary-exact grid: voxel 1/16 m, truncation 4 vox = 0.25 m,
    depth 1.78125 m everywhere -> the voxel layer at z = 2.03125 sits EXACTLY at sdf == -truncation (all values exact in f32)."""
    cam = (64.0, 64.0, 64.0, 48.0, 128, 96)
    depth = np.full((96, 128), 1.78125, np.float32)
    return cam, depth, np.eye(4, dtype=np.float32)


@pytest.mark.parametrize("skip", [0, 1])
def test_voxel_exactly_at_negative_truncation(oracle_mod, hip_lib, skip):
    kw = dict(voxel_size=0.0625, depth_interp_nearest=1, tsdf_skip_at_negative_truncation=skip)
    M, g, o = make_pair(oracle_mod, **kw)
    cam, depth, T = _edge_case_frame()
    g.integrate_depth(depth, T, cam); o.integrate_depth(depth, T, cam)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    # the layer of voxel centres at z = 2.03125 (block z 4, voxel z 0): observed iff the switch says "integrate"
    tm = tsdf_map(g, M)
    blk = tm[(0, 0, 4)]
    w_edge = blk["weight"].reshape(8, 8, 8)[:, :, 0]           # [x][y][z] order z + 8y + 64x
    d_edge = blk["distance"].reshape(8, 8, 8)[:, :, 0]
    if skip:
        assert (w_edge == 0).all()
    else:
        assert (w_edge == 1).all() and (d_edge == np.float32(-0.25)).all()
    # the layer in front of it (z = 1.96875, sdf = -0.1875) is integrated either way
    assert (tm[(0, 0, 3)]["weight"].reshape(8, 8, 8)[:, :, 7] == 1).all()


@pytest.mark.parametrize("before", [0, 1])
def test_max_weight_clamp_order(oracle_mod, hip_lib, before):
    kw = dict(max_weight=2.0, tsdf_weight_clamp_before_blend=before)
    M, g, o = make_pair(oracle_mod, **kw)
    sc = S.Scene()
    T = S.trajectory_pose(3)
    rng = np.random.default_rng(5)
    for k in range(5):                       # the same view five times with different noise: the weights saturate at 2
        d, _ = S.render(sc, T, H.SMALL_CAM, color=False, noise_sigma=0.01, rng=rng)
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 50
    test_max_weight_clamp_order.maps = getattr(test_max_weight_clamp_order, "maps", {})
    test_max_weight_clamp_order.maps[before] = tsdf_map(g, M)
    if len(test_max_weight_clamp_order.maps) == 2:
        a, b = test_max_weight_clamp_order.maps[0], test_max_weight_clamp_order.maps[1]
        assert all(np.array_equal(a[k]["weight"], b[k]["weight"]) for k in a)           # same weights ...
        assert any(not np.array_equal(a[k]["distance"], b[k]["distance"]) for k in a)   # ... different running averages


@pytest.mark.parametrize("thresh", [-1.0, 0.5, 8.0])
def test_colour_occlusion_threshold(oracle_mod, hip_lib, thresh):
    M, g, o = make_pair(oracle_mod, color_occlusion_threshold_vox=thresh)
    for d, rgb, T in H.frames(4, H.SMALL_CAM, color=True, stride=6):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(rgb, T, H.SMALL_CAM); o.integrate_color(rgb, T, H.SMALL_CAM)
    n, worst = compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    assert n > 20
    idx = g.block_indices(M.LAYER_COLOR)
    cb, _ = g.get_blocks(M.LAYER_COLOR, idx)
    test_colour_occlusion_threshold.n = getattr(test_colour_occlusion_threshold, "n", {})
    test_colour_occlusion_threshold.n[thresh] = int((cb["weight"] > 0).sum())
    if len(test_colour_occlusion_threshold.n) == 3:
        n_ = test_colour_occlusion_threshold.n
        assert n_[0.5] < n_[-1.0] < n_[8.0], n_           # a tighter occlusion test colours fewer voxels (-1 = truncation = 4 vox)


def numpy_propagation(sites, dom, max_sq):
    """Independent restatement (numpy, whole-array) of esdf_propagation = 1: synchronous 4-neighbour parent propagation, key
    (sq, dy, dx), cut-off max_sq, restricted to `dom`.  Returns sq (float, max_sq where no site is known)."""
    H_, W_ = sites.shape
    NONE = np.int64(2 ** 31 - 1)
    cur = np.where(sites, np.int64((64 << 7) | 64), NONE)
    for _ in range(4096):
        best = cur.copy()
        for ox, oy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
            nb = np.full_like(cur, NONE)
            # neighbour at (x + ox, y + oy)
            ys = slice(max(0, -oy), H_ - max(0, oy)); xs = slice(max(0, -ox), W_ - max(0, ox))
            ysn = slice(max(0, oy), H_ - max(0, -oy)); xsn = slice(max(0, ox), W_ - max(0, -ox))
            nb[ys, xs] = np.where(dom[ysn, xsn], cur[ysn, xsn], NONE)
            dx = (nb & 127) - 64 + ox; dy = ((nb >> 7) & 127) - 64 + oy
            sq = dx * dx + dy * dy
            ok = (nb != NONE) & (sq.astype(np.float32) <= np.float32(max_sq)) & (np.abs(dx) <= 63) & (np.abs(dy) <= 63)
            cand = np.where(ok, (sq << 14) | ((dy + 64) << 7) | (dx + 64), NONE)
            best = np.minimum(best, cand)
        best = np.where(dom, best, NONE)
        if np.array_equal(best, cur):
            break
        cur = best
    return np.where(cur == NONE, np.float32(max_sq), (cur >> 14).astype(np.float32))


@pytest.mark.parametrize("prop", [0, 1])
def test_esdf_exact_transform_vs_iterative_propagation(oracle_mod, hip_lib, prop):
    M, g, o = make_pair(oracle_mod, esdf_propagation=prop)
    fr = H.frames(6, H.SMALL_CAM, color=False, stride=13)
    for k, (d, rgb, T) in enumerate(fr):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        if k % 2 == 1:
            g.update_esdf(); o.update_esdf()
            ig, ag = g.esdf_slice_image(1000.0); io, ao = o.esdf_slice_image(1000.0)
            assert ig.shape == io.shape and np.array_equal(ag, ao) and np.abs(ig - io).max() <= TOL
    n, worst = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF,
                             fields_exact=("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site"))
    assert n > 10
    # independent check of the definition itself (numpy): dense site / domain images from the GPU's own ESDF layer
    idx = g.block_indices(M.LAYER_ESDF)
    eb, _ = g.get_blocks(M.LAYER_ESDF, idx)
    vz = int(np.floor(g.params.esdf_slice_height / g.params.voxel_size)) & 7
    x0, y0 = idx[:, 0].min(), idx[:, 1].min()
    W_, H_ = (idx[:, 0].max() - x0 + 1) * 8, (idx[:, 1].max() - y0 + 1) * 8
    sites = np.zeros((H_, W_), bool); dom = np.zeros((H_, W_), bool); sq = np.zeros((H_, W_), np.float32)
    for k, (bx, by, bz) in enumerate(idx.tolist()):
        pl = eb[k].reshape(8, 8, 8)[:, :, vz]                      # [x][y]
        ys, xs = slice((by - y0) * 8, (by - y0) * 8 + 8), slice((bx - x0) * 8, (bx - x0) * 8 + 8)
        sites[ys, xs] = pl["is_site"].T != 0; dom[ys, xs] = True; sq[ys, xs] = pl["squared_distance_vox"].T
    r = g.params.esdf_max_distance_m / g.params.voxel_size
    if prop == 1:
        ref = numpy_propagation(sites, dom, np.float32(r) * np.float32(r))
        assert np.array_equal(ref[dom], sq[dom])
    test_esdf_exact_transform_vs_iterative_propagation.sq = getattr(test_esdf_exact_transform_vs_iterative_propagation, "sq", {})
    test_esdf_exact_transform_vs_iterative_propagation.sq[prop] = (sq, dom)
    if len(test_esdf_exact_transform_vs_iterative_propagation.sq) == 2:
        (a, da), (b, db) = (test_esdf_exact_transform_vs_iterative_propagation.sq[k] for k in (0, 1))
        assert np.array_equal(da, db)
        assert (b[da] >= a[da]).all()                 # propagation can only over-estimate the exact distance ...
        assert ((b[da] - a[da]) > 0).mean() < 0.05    # ... and does so rarely (vector-propagation error + blocked paths)


@pytest.mark.parametrize("normal_rule", [0, 1])
@pytest.mark.parametrize("rule", [0, 1, 2])
def test_mesh_ambiguity_and_normal_rules(oracle_mod, hip_lib, rule, normal_rule):
    M, g, o = make_pair(oracle_mod, mesh_ambiguity_rule=rule, mesh_normal_rule=normal_rule)
    rng = np.random.default_rng(11)
    sc = S.Scene()
    for k in range(4):        # noisy depth: ambiguous cube configurations actually occur
        T = S.trajectory_pose(k * 8)
        d, rgb = S.render(sc, T, H.SMALL_CAM, color=True, noise_sigma=0.02, rng=rng)
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(rgb, T, H.SMALL_CAM); o.integrate_color(rgb, T, H.SMALL_CAM)
    g.update_color_mesh(); o.update_mesh()
    mg = g.mesh()
    nonempty = 0; ntri = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        mo = o.mesh_block(idx)
        a = mg[tuple(idx)]
        assert a["triangles"].shape == mo["triangles"].shape and np.array_equal(a["triangles"], mo["triangles"]), idx
        if len(mo["vertices"]):
            nonempty += 1; ntri += len(mo["triangles"])
            assert np.abs(a["vertices"] - mo["vertices"]).max() <= TOL
            assert np.abs(a["normals"] - mo["normals"]).max() <= 1e-3
    assert nonempty > 20
    test_mesh_ambiguity_and_normal_rules.ntri = getattr(test_mesh_ambiguity_and_normal_rules, "ntri", {})
    test_mesh_ambiguity_and_normal_rules.ntri[(rule, normal_rule)] = ntri
    nt = test_mesh_ambiguity_and_normal_rules.ntri
    if len(nt) == 6:
        assert nt[(0, 0)] == nt[(0, 1)] and nt[(2, 0)] == nt[(2, 1)]
        assert len({nt[(0, 0)], nt[(1, 0)], nt[(2, 0)]}) >= 2, nt     # the noisy surface does contain ambiguous faces


@pytest.mark.parametrize("thr", [(2.0, 0.5), (0.5, 0.25), (8.0, 1.5)])
def test_lidar_interpolation_thresholds(oracle_mod, hip_lib, thr):
    """[U] interpolateLidarImage acceptance thresholds (bilinear taps agree within a, nearest beam within b voxels of the ray)."""
    from isaac_ros_nvblox_amd import mapper as M
    lidar = (256, 16, 0.1, -np.deg2rad(15.0), np.deg2rad(15.0))
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=2,
              lidar_linear_interpolation_max_allowable_difference_vox=thr[0],
              lidar_nearest_interpolation_max_allowable_dist_to_ray_vox=thr[1])
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 16); o = oracle_mod.OracleMap(po)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    for i in range(2):
        T = S.lidar_pose(i * 7)
        img = S.render_lidar(sc, T, lidar, max_range=60.0)
        g.integrate_lidar_depth(img, T, lidar); o.integrate_lidar_depth(img, T, lidar)
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 500
    bg, _ = g.get_blocks(M.LAYER_TSDF, g.block_indices(M.LAYER_TSDF))
    test_lidar_interpolation_thresholds.n = getattr(test_lidar_interpolation_thresholds, "n", {})
    test_lidar_interpolation_thresholds.n[thr] = int((bg["weight"] > 0).sum())
    if len(test_lidar_interpolation_thresholds.n) == 3:
        n_ = test_lidar_interpolation_thresholds.n
        assert n_[(0.5, 0.25)] < n_[(2.0, 0.5)] < n_[(8.0, 1.5)], n_


@pytest.mark.parametrize("kw", [dict(depth_interp_nearest=1), dict(esdf_site_rule=1), dict(depth_interp_nearest=1, esdf_site_rule=1)])
def test_depth_sampling_and_site_rule_switches(oracle_mod, hip_lib, kw):
    """The two switches round 1 already had: nearest vs bilinear-with-validity depth sampling; ESDF site rule |d| <= s with / without 'inside'."""
    M, g, o = make_pair(oracle_mod, **kw)
    for d, rgb, T in H.frames(4, H.SMALL_CAM, color=False, stride=9):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    g.update_esdf(); o.update_esdf()
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    n, _ = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF,
                         fields_exact=("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site"))
    assert n > 10


def test_plain_instantiation_equals_the_general_kernel(oracle_mod, hip_lib):
    """k_integrate_tsdf has a compile-time folded instantiation for the default configuration (frame_is_plain, nvbx_internal.h).  An
    invalid-depth decay factor of exactly 1.0 selects the general kernel without changing any value (weight * 1.0), so the two maps must be
    bit-identical -- camera (with holes in the depth image, so the decay branch really runs) and LiDAR."""
    from isaac_ros_nvblox_amd import mapper as M
    gp = M.Mapper(M.default_params(), block_capacity=1 << 14)
    gg = M.Mapper(M.default_params(invalid_depth_decay_factor=1.0), block_capacity=1 << 14)
    for k, (d, rgb, T) in enumerate(H.frames(4, H.SMALL_CAM, color=False, stride=9)):
        d = d.copy(); d[10 + 5 * k:30 + 5 * k, 20:60] = 0.0
        gp.integrate_depth(d, T, H.SMALL_CAM); gg.integrate_depth(d, T, H.SMALL_CAM)
    a, b = tsdf_map(gp, M), tsdf_map(gg, M)
    assert len(a) > 100 and not maps_differ(a, b)
    lidar = (256, 16, 0.1, -np.deg2rad(15.0), np.deg2rad(15.0))
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=30.0, raycast_subsampling_factor=2)
    gp = M.Mapper(M.default_params(**kw), block_capacity=1 << 16)
    gg = M.Mapper(M.default_params(invalid_depth_decay_factor=1.0, **kw), block_capacity=1 << 16)
    sc = S.LidarScene(n_boxes=12, extent=40.0)
    for i in range(2):
        T = S.lidar_pose(i * 7)
        img = S.render_lidar(sc, T, lidar, max_range=40.0)
        gp.integrate_lidar_depth(img, T, lidar); gg.integrate_lidar_depth(img, T, lidar)
    a, b = tsdf_map(gp, M), tsdf_map(gg, M)
    assert len(a) > 1000 and not maps_differ(a, b)
