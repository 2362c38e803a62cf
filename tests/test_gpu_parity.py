"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): block-index sets bit-exact; TSDF / ESDF voxel values within 1e-4 (we observe 0);
colour within +-1 LSB and weights within 1e-4; mesh triangle sets equal, vertices within 1e-4.
"""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu

TOL = 1e-4


def make_pair(oracle_mod, **kw):
    from isaac_ros_nvblox_amd import mapper as M
    pg = M.default_params(**kw)
    po = H.copy_params(pg, oracle_mod.OrcParams)
    return M, M.Mapper(pg, block_capacity=1 << 14), oracle_mod.OracleMap(po)


def compare_layer(M, g, o, layer_g, layer_o, fields_exact=(), fields_tol=(), lsb_fields=()):
    ig = g.block_indices(layer_g); io = o.block_indices(layer_o)
    assert H.idx_set(ig) == H.idx_set(io), "block index sets differ: only-gpu %s only-oracle %s" % (
        sorted(H.idx_set(ig) - H.idx_set(io))[:5], sorted(H.idx_set(io) - H.idx_set(ig))[:5])
    assert np.array_equal(ig, io)            # both sorted lexicographically
    bg, found = g.get_blocks(layer_g, ig)
    assert found.all()
    worst = 0.0
    for k, idx in enumerate(io):
        bo = o.get_block(layer_o, idx)
        for f in fields_exact:
            assert np.array_equal(bg[k][f], bo[f]), (f, idx)
        for f in fields_tol:
            d = np.abs(bg[k][f].astype(np.float64) - bo[f].astype(np.float64)).max()
            worst = max(worst, d)
            assert d <= TOL, (f, idx, d)
        for f in lsb_fields:
            d = np.abs(bg[k][f].astype(np.int32) - bo[f].astype(np.int32)).max()
            assert d <= 1, (f, idx, d)
    return len(io), worst


@pytest.mark.parametrize("weighting_mode", [0, 1, 2, 3, 4, 5])     # all six WeightingFunctionType values (mapper_initialization.cpp:31-42)
def test_tsdf_parity_small(oracle_mod, hip_lib, weighting_mode):
    M, g, o = make_pair(oracle_mod, weighting_mode=weighting_mode)
    for d, rgb, T in H.frames(6, H.SMALL_CAM, color=False, stride=7):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 100
    assert g.counters()["capacity_overflow"] == 0


def test_tsdf_parity_full_res(oracle_mod, hip_lib):
    """BASELINE.json configs[1] shape: 640x480 @ 0.05 m, fuser.yaml parameters, 4 frames."""
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(4, S.REPLICA_LIKE_CAM, color=False, stride=10):
        g.integrate_depth(d, T, S.REPLICA_LIKE_CAM); o.integrate_depth(d, T, S.REPLICA_LIKE_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
        assert g.counters()["tsdf_blocks_in_view"] == len(o.last_view())
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 300


def test_depth_u16mm_fused_conversion(oracle_mod, hip_lib):
    """uint16 millimetre depth (image_conversions_thrust.cu:39-45 DivideBy1000) fused into the integrator read."""
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(2, H.SMALL_CAM, color=False, stride=9):
        mm = np.round(d * 1000.0).astype(np.uint16)
        g.integrate_depth(mm, T, H.SMALL_CAM)
        o.integrate_depth(mm.astype(np.float32) * np.float32(1.0 / 1000.0), T, H.SMALL_CAM)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))


def test_invalid_depth_decay_parity(oracle_mod, hip_lib):
    """projective_tsdf_integrator_invalid_depth_decay_factor (mapper_initialization.cpp:294-300; 0.8 in nvblox_dynamics.yaml:11):
    voxels that project onto invalid depth lose weight."""
    M, g, o = make_pair(oracle_mod, invalid_depth_decay_factor=0.8)
    fr = H.frames(4, H.SMALL_CAM, color=False, stride=5)
    for k, (d, rgb, T) in enumerate(fr):
        d = d.copy()
        if k >= 2:
            d[30:80, 40:110] = 0.0          # a hole of invalid depth over previously observed space
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    # the decay actually happened: some weights are 2 * 0.8^2 = 1.28 (two clean frames, two decays)
    bg, _ = g.get_blocks(M.LAYER_TSDF, g.block_indices(M.LAYER_TSDF))
    assert (np.abs(bg["weight"] - np.float32(2.0 * 0.8 * 0.8)) < 1e-5).sum() > 100


@pytest.mark.parametrize("bounds", [(1, (0, 0, 0.3), (0, 0, 1.9)), (2, (-1.0, -2.0, 0.0), (2.5, 1.0, 2.2))])
def test_workspace_bounds_parity(oracle_mod, hip_lib, bounds):
    """workspace_bounds_type height_bounds / bounding_box (mapper_initialization.cpp:337-358, nvblox_base.yaml:88-94):
    blocks outside the bounds are neither allocated nor integrated."""
    from isaac_ros_nvblox_amd import mapper as M
    btype, lo, hi = bounds
    pg = M.default_params(); pg.workspace_bounds_type = btype
    pg.workspace_bounds_min_corner_m[:] = lo; pg.workspace_bounds_max_corner_m[:] = hi
    po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 14); o = oracle_mod.OracleMap(po)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=False, stride=13):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    idx = g.block_indices(M.LAYER_TSDF)
    assert n > 30
    assert ((idx[:, 2] + 1) * 0.4 > lo[2]).all() and (idx[:, 2] * 0.4 < hi[2]).all()
    if btype == 2:
        assert ((idx[:, 0] + 1) * 0.4 > lo[0]).all() and (idx[:, 0] * 0.4 < hi[0]).all()
    # and the unbounded map is strictly larger
    g2 = M.Mapper(M.default_params(), block_capacity=1 << 14)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=False, stride=13):
        g2.integrate_depth(d, T, H.SMALL_CAM)
    assert g2.num_blocks(M.LAYER_TSDF) > n


def test_clear_tsdf_inside_shapes_parity(oracle_mod, hip_lib):
    """Mapper::clearTsdfInsideShapes (nvblox_node.cpp:1834): sphere + box, then ESDF / mesh follow."""
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=False, stride=9):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    g.update_esdf(); o.update_esdf(); g.update_color_mesh(); o.update_mesh()
    shapes = [("sphere", (1.5, 1.0, 0.6), 0.8), ("aabb", (-3.1, -2.6, -0.1), (-1.0, 0.0, 1.0))]
    g.clear_tsdf_inside_shapes(shapes); n_cleared = o.clear_tsdf_inside_shapes(shapes)
    assert n_cleared > 1000
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    g.update_esdf(); o.update_esdf()
    sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
    g.update_color_mesh(); o.update_mesh()
    mg = g.mesh()
    assert len(mg) > 0
    for idx, mb in mg.items():
        mo = o.mesh_block(np.array(idx, np.int32))
        assert np.array_equal(mb["triangles"], mo["triangles"])


def test_depth_preprocessing_parity(oracle_mod, hip_lib):
    """do_depth_preprocessing / depth_preprocessing_num_dilations (mapper_initialization.cpp:238-243): invalid-depth regions are
    dilated before integration; f32 and u16-mm inputs."""
    M, g, o = make_pair(oracle_mod, do_depth_preprocessing=1, depth_preprocessing_num_dilations=3)
    for k, (d, rgb, T) in enumerate(H.frames(3, H.SMALL_CAM, color=False, stride=7)):
        d = d.copy(); d[20:40, 50:90] = 0.0; d[100:104, 10:14] = 0.0
        if k == 2:
            mm = np.round(d * 1000.0).astype(np.uint16)
            g.integrate_depth(mm, T, H.SMALL_CAM); o.integrate_depth(mm.astype(np.float32) * np.float32(1.0 / 1000.0), T, H.SMALL_CAM)
        else:
            g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, worst = compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 100


def test_esdf_parity(oracle_mod, hip_lib):
    M, g, o = make_pair(oracle_mod)
    fr = H.frames(8, H.SMALL_CAM, color=False, stride=11)
    for k, (d, rgb, T) in enumerate(fr):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        if k % 2 == 1:      # incremental updates: window logic must reproduce the oracle's full recompute
            g.update_esdf(); o.update_esdf()
            ig, ag = g.esdf_slice_image(1000.0); io, ao = o.esdf_slice_image(1000.0)
            assert ig.shape == io.shape and np.array_equal(ag, ao)
            assert np.abs(ig - io).max() <= TOL
    n, worst = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF,
                             fields_exact=("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site"))
    assert n > 10
    c = g.counters()
    assert c["esdf_blocks_swept"] > 0 and c["capacity_overflow"] == 0


def test_color_parity(oracle_mod, hip_lib):
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(5, H.SMALL_CAM, color=True, stride=6):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(rgb, T, H.SMALL_CAM); o.integrate_color(rgb, T, H.SMALL_CAM)
        sg, so = g.synthetic_depth(), o.synthetic_depth()
        assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
        assert H.idx_set(g.last_color_view()) == H.idx_set(o.last_color_view())
    n, worst = compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    assert n > 20


def test_color_bgra8_fused_conversion(oracle_mod, hip_lib):
    """bgra8 colour input (image_conversions.cpp:170-176, ToRgba<Bgra> image_conversions_thrust.cu:60-65) fused into the fetch."""
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=True, stride=6):
        rgb = rgb.copy(); rgb[..., 1] = (rgb[..., 1] // 2); rgb[..., 2] = 255 - rgb[..., 2]     # make the three channels distinct
        bgra = np.concatenate([rgb[..., ::-1], np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(bgra, T, H.SMALL_CAM); o.integrate_color(rgb, T, H.SMALL_CAM)
    n, worst = compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    assert n > 20


def test_mesh_parity(oracle_mod, hip_lib):
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(4, H.SMALL_CAM, color=True, stride=8):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(rgb, T, H.SMALL_CAM); o.integrate_color(rgb, T, H.SMALL_CAM)
    g.update_color_mesh(); o.update_mesh()
    mg = g.mesh()
    nonempty = 0
    for idx in o.block_indices(oracle_mod.L_TSDF):
        mo = o.mesh_block(idx)
        assert tuple(idx) in mg, idx
        a = mg[tuple(idx)]
        assert a["triangles"].shape == mo["triangles"].shape and np.array_equal(a["triangles"], mo["triangles"]), idx
        assert a["vertices"].shape == mo["vertices"].shape
        if len(mo["vertices"]):
            nonempty += 1
            assert np.abs(a["vertices"] - mo["vertices"]).max() <= TOL
            assert np.abs(a["normals"] - mo["normals"]).max() <= 1e-3
            assert np.abs(a["colors"].astype(int) - mo["colors"].astype(int)).max() <= 1
    assert nonempty > 20
    # full-layer update gives the same mesh
    g.update_color_mesh(full=True)
    mg2 = g.mesh()
    assert set(mg2.keys()) == set(mg.keys())
    for k in mg:
        assert np.array_equal(mg[k]["triangles"], mg2[k]["triangles"]) and np.array_equal(mg[k]["vertices"], mg2[k]["vertices"])


def test_kat_esdf_dense_grid_gpu(oracle_mod, hip_lib):
    """Port of nvblox_ros/test/unit_tests/test_esdf_and_gradient_conversions.cpp:110-157 against the HIP path."""
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=256)
    vox = np.zeros(512, M.ESDF_DT)
    for x in range(8):
        for y in range(8):
            for z in range(8):
                vox[z + 8 * y + 64 * x]["squared_distance_vox"] = oracle_mod.lib().orc_index_hash(x, y, z) % 1000
                vox[z + 8 * y + 64 * x]["observed"] = 1
    g.set_block(M.LAYER_ESDF, (0, 0, 0), vox)
    # aabb of allocated blocks = [0, 0.4]^3 -> voxels 0..8 inclusive (the reference test iterates min..max inclusive)
    grid = g.esdf_dense_grid((0, 0, 0), (9, 9, 9), -1000.0)
    for x in range(9):
        for y in range(9):
            for z in range(9):
                if x < 8 and y < 8 and z < 8:
                    want = np.float32(0.05) * np.sqrt(np.float32(oracle_mod.lib().orc_index_hash(x, y, z) % 1000))
                    assert abs(grid[x, y, z] - want) <= 1e-6
                else:
                    assert abs(grid[x, y, z] - (-1000.0)) <= 1e-6
    back = g.get_block(M.LAYER_ESDF, (0, 0, 0))
    assert np.array_equal(back["squared_distance_vox"], vox["squared_distance_vox"])


def test_slice_consumers(oracle_mod, hip_lib):
    """Slice -> occupancy grid / point cloud (esdf_slice_conversions.cu:33-73, nvblox_node.cpp:917-919)."""
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=False, stride=13):
        g.integrate_depth(d, T, H.SMALL_CAM)
    g.update_esdf()
    img, aabb = g.esdf_slice_image_device(1000.0)
    host = img.cpu().numpy()
    occ = g.occupancy_grid_from_slice(img, 1000.0).cpu().numpy()
    assert ((occ == -1) == (np.abs(host - 1000.0) < 1e-2)).all()
    assert ((occ == 100) == ((host <= 0) & (np.abs(host - 1000.0) >= 1e-2))).all()
    pts = g.pointcloud_from_slice(img, aabb, 0.09, 1000.0).cpu().numpy()
    known = np.abs(host - 1000.0) >= 1e-2
    assert len(pts) == known.sum()
    rows, cols = np.nonzero(known)
    want = np.stack([aabb[0] + np.float32(0.05) * cols.astype(np.float32), aabb[1] + np.float32(0.05) * rows.astype(np.float32),
                     np.full(len(rows), 0.09, np.float32), host[known]], 1)
    assert np.allclose(np.array(sorted(map(tuple, pts.tolist()))), np.array(sorted(map(tuple, want.tolist()))), atol=1e-6)


def test_decay_and_clear_parity(oracle_mod, hip_lib):
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.5, tsdf_decayed_weight_threshold=0.3)
    fr = H.frames(4, H.SMALL_CAM, color=False, stride=25)
    for d, rgb, T in fr:
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    g.update_esdf(); o.update_esdf()
    for _ in range(3):
        g.decay_tsdf(True); o.decay_tsdf(True)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    g.update_esdf(); o.update_esdf()
    ig, ag = g.esdf_slice_image(1000.0); io, ao = o.esdf_slice_image(1000.0)
    assert ig.shape == io.shape and np.abs(ig - io).max() <= TOL
    # integrate again after deallocation (slots are recycled, hash was rebuilt on device)
    d, rgb, T = fr[1]
    g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    c = (float(T[0, 3]), float(T[1, 3]), 1.0)
    g.clear_outside_radius(c, 2.0); o.clear_outside_radius(c, 2.0)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))


def compare_occupancy(M, g, o, oracle_mod):
    ig = g.block_indices(M.LAYER_OCCUPANCY); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io), (len(ig), len(io))
    bg, found = g.get_blocks(M.LAYER_OCCUPANCY, ig)
    assert found.all()
    for k, idx in enumerate(io):
        assert np.array_equal(bg[k]["log_odds"], o.get_block(oracle_mod.L_TSDF, idx)["distance"]), idx     # adds and compares only: bit-exact
    return len(io), bg


def test_occupancy_mapper_parity(oracle_mod, hip_lib):
    """Mapper(voxel_size, memory_type, ProjectiveLayerType::kOccupancy): mapping_type static_occupancy (nvblox_base.yaml:9) and the
    dynamic mapper of the segmentation configuration (specializations/nvblox_segmentation.yaml:9-22): log-odds integration,
    ESDF from occupancy, decayOccupancyAllVoxels with deallocation, clearing; colour and mesh are no-ops."""
    occ = dict(projective_layer_type=1, free_region_occupancy_probability=0.3, occupied_region_occupancy_probability=0.9,
               unobserved_region_occupancy_probability=0.35, occupied_region_half_width_m=0.2,
               free_region_decay_probability=0.55, occupied_region_decay_probability=0.30, max_integration_distance_m=5.0)
    M, g, o = make_pair(oracle_mod, **occ)
    fr = H.frames(5, H.SMALL_CAM, color=True, stride=9)
    for d, rgb, T in fr:
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.integrate_color(rgb, T, H.SMALL_CAM)                                   # no-op on an occupancy mapper
        assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
    n, bg = compare_occupancy(M, g, o, oracle_mod)
    lo = bg["log_odds"]
    assert n > 100 and (lo > 0).sum() > 1000 and (lo < 0).sum() > 10000 and np.abs(lo).max() <= 10.0
    assert g.num_blocks(M.LAYER_TSDF) == 0 and g.num_blocks(M.LAYER_COLOR) == 0
    g.update_color_mesh(); assert len(g.mesh()) == 0
    g.update_esdf(); o.update_esdf()
    sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
    assert sg.shape == so.shape and np.array_equal(sg, so) and (sg <= 0).sum() > 50 and (np.abs(sg - 1000.0) > 1).mean() > 0.2
    n_before = n
    for k in range(14):
        g.decay_occupancy(); o.decay_occupancy()
        if k % 4 == 3:
            n, _ = compare_occupancy(M, g, o, oracle_mod)
            g.update_esdf(); o.update_esdf()
            sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
            assert sg.shape == so.shape and np.array_equal(sg, so)
    assert n < n_before                                                          # fully decayed blocks were deallocated
    d, rgb, T = fr[2]
    g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    c = (float(T[0, 3]), float(T[1, 3]), 1.0)
    g.clear_outside_radius(c, 2.5); o.clear_outside_radius(c, 2.5)
    compare_occupancy(M, g, o, oracle_mod)
    g.update_esdf(); o.update_esdf()
    sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
    assert sg.shape == so.shape and np.array_equal(sg, so)
    with pytest.raises(RuntimeError):
        g.decay_tsdf(True)                                                       # wrong decay for this layer type
    assert g.counters()["capacity_overflow"] == 0


def test_occupancy_map_file_round_trip(hip_lib, tmp_path):
    from isaac_ros_nvblox_amd import mapper as M
    a = M.Mapper(M.default_params(projective_layer_type=1), block_capacity=1 << 12)
    for d, _, T in H.frames(2, H.SMALL_CAM, color=False, stride=9):
        a.integrate_depth(d, T, H.SMALL_CAM)
    a.update_esdf()
    p = tmp_path / "occ.nvbxmap"; a.save_map(p)
    b = M.Mapper(M.default_params(projective_layer_type=1), block_capacity=1 << 12); b.load_map(p)
    ia = a.block_indices(M.LAYER_OCCUPANCY)
    assert len(ia) > 50 and np.array_equal(ia, b.block_indices(M.LAYER_OCCUPANCY))
    assert a.get_blocks(M.LAYER_OCCUPANCY, ia)[0].tobytes() == b.get_blocks(M.LAYER_OCCUPANCY, ia)[0].tobytes()
    t = M.Mapper(M.default_params(), block_capacity=1 << 12)
    with pytest.raises(RuntimeError):
        t.load_map(p)                                                            # a TSDF mapper cannot hold an occupancy layer


def test_mask_split_and_human_mapping_parity(oracle_mod, hip_lib):
    """MultiMapper::integrateDepth(depth, mask, T_L_CD, T_CM_CD, depth_cam, mask_cam) (nvblox_node.cpp:1018-1060) as the
    human-mapping configuration runs it (specializations/nvblox_segmentation.yaml): the depth image is split by the mask
    (mask camera offset from the depth camera, occlusion test), the unmasked part goes into the static TSDF mapper, the
    masked part into the occupancy mapper.  Split images bit-exact; both maps equal to the oracle fed the same way."""
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    mask_cam = (85.0, 85.0, 79.5, 59.5, 160, 120)
    T_CM_CD = np.eye(4, dtype=np.float32); T_CM_CD[0, 3] = 0.06; T_CM_CD[1, 3] = -0.01      # a colour camera 6 cm beside the depth camera
    occ = dict(projective_layer_type=1, free_region_occupancy_probability=0.3, occupied_region_occupancy_probability=0.9,
               unobserved_region_occupancy_probability=0.35, occupied_region_half_width_m=0.2, max_integration_distance_m=5.0)
    _, gs, os_ = make_pair(oracle_mod)
    _, gd, od = make_pair(oracle_mod, **occ)
    for k, (d, rgb, T) in enumerate(H.frames(4, cam, color=True, stride=7)):
        d = d.copy(); d[:6, :] = 0.0
        mask = np.zeros((120, 160), np.uint8)
        mask[30:100, 50 + 5 * k:95 + 5 * k] = 1 + k                                # the "person": any non-zero value
        un_g, ma_g, ov = gs.split_depth_by_mask(d, mask, T_CM_CD, cam, mask_cam, 0.25, overlay=True)
        un_o, ma_o = oracle_mod.split_depth_by_mask(d, mask, T_CM_CD, cam, mask_cam, 0.25)
        un_g, ma_g = un_g.cpu().numpy(), ma_g.cpu().numpy()
        assert np.array_equal(un_g, un_o) and np.array_equal(ma_g, ma_o)
        n_masked = int((ma_g > 0).sum())
        assert 1000 < n_masked < 5000 and ((un_g > 0) & (ma_g > 0)).sum() == 0 and np.array_equal((un_g > 0) | (ma_g > 0), d > 0)
        assert (ov.cpu().numpy()[..., 0] == 255).sum() >= n_masked
        gs.integrate_depth(un_g, T, cam); os_.integrate_depth(un_o, T, cam)         # background: static TSDF
        gd.integrate_depth(ma_g, T, cam); od.integrate_depth(ma_o, T, cam)         # foreground: occupancy
        cu, cm = gs.split_color_by_mask(rgb, mask)
        cu = cu.cpu().numpy(); cm = cm.cpu().numpy()
        assert np.array_equal(cu, np.where(mask[..., None] != 0, 0, rgb)) and np.array_equal(cm, np.where(mask[..., None] != 0, rgb, 0))
    compare_layer(M, gs, os_, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    n, _ = compare_occupancy(M, gd, od, oracle_mod)
    assert n > 10
    gs.update_esdf(); os_.update_esdf(); gd.update_esdf(); od.update_esdf()
    for g, o in ((gs, os_), (gd, od)):
        sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
        assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
    comb, _ = gs.esdf_slice_image_combined(gd)                                     # the combined costmap of nvblox_node.cpp:836-840
    assert (comb < 999.0).sum() >= (gs.esdf_slice_image()[0] < 999.0).sum()
    # occlusion test: a background pixel far behind the person that lands on the same mask pixel is NOT masked
    d2 = np.full((120, 160), 4.0, np.float32); d2[:, 80:] = 1.0                    # near surface on the right half
    m2 = np.ones((120, 160), np.uint8)
    Tshift = np.eye(4, dtype=np.float32); Tshift[0, 3] = -0.5                      # strong parallax: far pixels slide onto near ones' mask pixels
    un2, ma2 = gs.split_depth_by_mask(d2, m2, Tshift, cam, cam, 0.25)
    un2o, ma2o = oracle_mod.split_depth_by_mask(d2, m2, Tshift, cam, cam, 0.25)
    assert np.array_equal(un2.cpu().numpy(), un2o) and np.array_equal(ma2.cpu().numpy(), ma2o)
    assert ((un2o > 0) & (d2 == 4.0)).sum() > 50                                   # occluded far pixels stay in the background image


def compare_esdf3(M, g, o, oracle_mod):
    n, _ = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF,
                         fields_exact=("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site"))
    return n


def test_esdf_3d_parity(oracle_mod, hip_lib):
    """EsdfMode::k3D (esdf_mode "3d", node_params.hpp:90): every voxel of every updated block; the whole ESDF layer (squared
    distances, parent directions, flags) bit-exact against the oracle over incremental updates, decay with deallocation and
    radius clearing; the distances are the true 3-D Euclidean distance transform (brute force on a sub-volume); slice and dense
    query read the same layer."""
    M, g, o = make_pair(oracle_mod, esdf_mode=1, esdf_max_distance_m=1.0, tsdf_decay_factor=0.5, tsdf_decayed_weight_threshold=0.3,
                        max_integration_distance_m=4.0)
    fr = H.frames(5, H.SMALL_CAM, color=False, stride=8)
    for k, (d, rgb, T) in enumerate(fr):
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        if k in (1, 4):
            g.update_esdf(); o.update_esdf()                             # incremental: the second update re-computes a window only
            n = compare_esdf3(M, g, o, oracle_mod)
    assert n > 150 and H.idx_set(g.block_indices(M.LAYER_ESDF)) == H.idx_set(g.block_indices(M.LAYER_TSDF))
    # brute force 3-D EDT on the GPU's own sites, a 5 x 5 x 3 block sub-volume in the middle of the map
    idx = g.block_indices(M.LAYER_ESDF)
    b, _ = g.get_blocks(M.LAYER_ESDF, idx)
    lo = idx.min(0); hi = idx.max(0)
    dims = (hi - lo + 1) * 8
    site = np.zeros(dims, bool); sq = np.full(dims, -1.0, np.float32)
    for k, i in enumerate(idx):
        s = (i - lo) * 8
        blk = b[k].reshape(8, 8, 8)                                      # [x][y][z]
        site[s[0]:s[0] + 8, s[1]:s[1] + 8, s[2]:s[2] + 8] = blk["is_site"].astype(bool)
        sq[s[0]:s[0] + 8, s[1]:s[1] + 8, s[2]:s[2] + 8] = blk["squared_distance_vox"]
    pts = np.argwhere(site)
    assert len(pts) > 2000
    c = dims // 2
    sub = np.stack(np.meshgrid(np.arange(c[0] - 20, c[0] + 20), np.arange(c[1] - 20, c[1] + 20), np.arange(max(c[2] - 12, 0), min(c[2] + 12, dims[2])),
                               indexing="ij"), -1).reshape(-1, 3)
    sub = sub[sq[sub[:, 0], sub[:, 1], sub[:, 2]] >= 0.0]                 # voxels of allocated blocks
    max_sq = np.float32((np.float32(1.0) / np.float32(0.05)) ** 2)
    best = np.full(len(sub), np.inf)
    for s0 in range(0, len(pts), 4000):
        d2 = ((sub[:, None, :] - pts[None, s0:s0 + 4000, :]) ** 2).sum(-1)
        best = np.minimum(best, d2.min(1))
    want = np.where(best <= max_sq, best, max_sq).astype(np.float32)
    assert len(sub) > 10000 and np.array_equal(sq[sub[:, 0], sub[:, 1], sub[:, 2]], want)
    # slice + dense query read the 3-D layer
    sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL and np.allclose(ag, ao)
    mn = (np.median(idx, axis=0).astype(np.int32) * 8 - np.array([20, 18, 10])).astype(np.int32); size = np.array([40, 36, 20], np.int32)
    dg = g.esdf_dense_grid(mn, size, 1000.0); do_ = o.esdf_dense_grid(mn, size, 1000.0)
    assert np.array_equal(dg, do_) and (dg < 999.0).mean() > 0.1
    # deallocation: decay until blocks die, radius clearing; ESDF follows
    for _ in range(3):
        g.decay_tsdf(True); o.decay_tsdf(True)
    g.update_esdf(); o.update_esdf()
    compare_esdf3(M, g, o, oracle_mod)
    d, rgb, T = fr[0]
    g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    cpos = (float(T[0, 3]), float(T[1, 3]), 1.0)
    g.clear_outside_radius(cpos, 2.2); o.clear_outside_radius(cpos, 2.2)
    g.update_esdf(); o.update_esdf()
    compare_esdf3(M, g, o, oracle_mod)
    assert g.counters()["capacity_overflow"] == 0


@pytest.mark.parametrize("cam", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_dynamic_mapping_parity(oracle_mod, hip_lib, cam):
    """MappingType::kDynamic (nvblox_dynamics.yaml): the static mapper carries a freespace layer (projective_layer_type 2); depth
    pixels whose points fall into high-confidence freespace are dynamic; the mask is cleaned of small components, splits the
    depth image, and the dynamic part feeds an occupancy mapper.  Freespace voxels (timestamps, durations, flags), dynamic masks,
    cleaned masks and both maps are bit-exact against the oracle."""
    from isaac_ros_nvblox_amd import mapper as M
    fs = dict(projective_layer_type=2, max_integration_distance_m=5.0, invalid_depth_decay_factor=0.8, max_tsdf_distance_for_occupancy_m=0.15,
              max_unobserved_to_keep_consecutive_occupancy_ms=200, min_duration_since_occupied_for_freespace_ms=250,
              min_consecutive_occupancy_duration_for_reset_ms=600, check_neighborhood=1, initialize_to_high_confidence_freespace=0)
    occ = dict(projective_layer_type=1, free_region_occupancy_probability=0.2, occupied_region_occupancy_probability=0.9,
               unobserved_region_occupancy_probability=0.35, occupied_region_half_width_m=0.15, max_integration_distance_m=5.0)
    _, gs, os_ = make_pair(oracle_mod, **fs)
    _, gd, od = make_pair(oracle_mod, **occ)
    static_scene = S.Scene()
    T = S.trajectory_pose(0, 200)
    eye = np.eye(4, dtype=np.float32)

    def frame(scene, t_ms, pose):
        d, _ = S.render(scene, pose, cam, color=False)
        mg = gs.detect_dynamics(d, pose, cam, 5.0); mo = os_.detect_dynamics(d, pose, cam, 5.0)
        assert np.array_equal(mg.cpu().numpy(), mo)
        cg = gs.remove_small_components(mg, 40); co = oracle_mod.remove_small_components(mo, 40)
        assert np.array_equal(cg.cpu().numpy(), co)
        un_g, ma_g = gs.split_depth_by_mask(d, cg, eye, cam, cam, 0.25)
        un_o, ma_o = oracle_mod.split_depth_by_mask(d, co, eye, cam, cam, 0.25)
        assert np.array_equal(un_g.cpu().numpy(), un_o) and np.array_equal(ma_g.cpu().numpy(), ma_o)
        gs.set_time_ms(t_ms); os_.set_time_ms(t_ms)
        gs.integrate_depth(un_g, pose, cam); os_.integrate_depth(un_o, pose, cam)
        gd.integrate_depth(ma_g, pose, cam); od.integrate_depth(ma_o, pose, cam)
        return mo, co

    def compare_freespace():
        ig = gs.block_indices(M.LAYER_FREESPACE); io = os_.block_indices(oracle_mod.L_FREESPACE)
        assert np.array_equal(ig, io) and len(io) > 50
        bg, found = gs.get_blocks(M.LAYER_FREESPACE, ig)
        assert found.all()
        n_free = 0
        for k, idx in enumerate(io):
            bo = os_.get_block(oracle_mod.L_FREESPACE, idx)
            for f in ("last_occupied_timestamp_ms", "consecutive_occupancy_duration_ms", "is_high_confidence_freespace", "initialized"):
                assert np.array_equal(bg[k][f], bo[f]), (f, idx)
            n_free += int(bo["is_high_confidence_freespace"].sum())
        return n_free

    # 1. the static room, observed for 0.9 s at 10 Hz from a slowly turning camera: free voxels become high-confidence freespace
    t = 0
    for k in range(10):
        mo, _ = frame(static_scene, t, S.trajectory_pose(k, 200)); t += 100
        if k < 3:
            assert mo.sum() == 0                      # nothing can be dynamic before any freespace exists
    n_free = compare_freespace()
    assert n_free > 20000
    # 2. an object appears in the middle of the room: its pixels are dynamic, the mask survives the clean-up, the static TSDF stays clean
    moving = S.Scene(box_min=(1.6, -0.3, 0.0), box_max=(2.0, 0.3, 1.3))
    n_dyn = []
    for k in range(5):
        mo, co = frame(moving, t, S.trajectory_pose(9, 200)); t += 100
        n_dyn.append(int(co.sum()))
    assert min(n_dyn) > 150, n_dyn
    compare_freespace()
    compare_layer(M, gs, os_, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    n, _ = compare_occupancy(M, gd, od, oracle_mod)
    assert n > 3
    # 3. the object stays for a second: the voxels it occupies reach the reset duration and stop being freespace -> it turns static
    #    (frames integrate the FULL depth into the static mapper here, as a long-standing object would be)
    d, _ = S.render(moving, S.trajectory_pose(9, 200), cam, color=False)
    for k in range(12):
        gs.set_time_ms(t); os_.set_time_ms(t)
        gs.integrate_depth(d, S.trajectory_pose(9, 200), cam); os_.integrate_depth(d, S.trajectory_pose(9, 200), cam); t += 100
    compare_freespace()
    mg = gs.detect_dynamics(d, S.trajectory_pose(9, 200), cam, 5.0).cpu().numpy(); mo = os_.detect_dynamics(d, S.trajectory_pose(9, 200), cam, 5.0)
    assert np.array_equal(mg, mo) and mo.sum() < 0.2 * n_dyn[0]
    # connected components on a random blob mask (many sizes, touching diagonally)
    rng = np.random.default_rng(5)
    mk = (rng.random((120, 160)) < 0.42).astype(np.uint8)
    for thr in (2, 9, 60, 400):
        assert np.array_equal(gs.remove_small_components(mk, thr).cpu().numpy(), oracle_mod.remove_small_components(mk, thr)), thr
