"""Round-2 regression tests (HIP path vs oracle): API orderings and edge cases that round 1's tests did not reach."""
import struct

import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_gpu_parity import TOL, compare_layer, make_pair

pytestmark = pytest.mark.gpu

ESDF_FIELDS = ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site")


def test_two_esdf_updates_back_to_back(oracle_mod, hip_lib):
    """update_esdf(); update_esdf(); with nothing in between: the held-back distance transform of the first must run before the second
    overwrites its arguments (nvbx_update_esdf scheduling note, include/nvblox_hip.h)."""
    M, g, o = make_pair(oracle_mod)
    fr = H.frames(4, H.SMALL_CAM, color=False, stride=11)
    for d, rgb, T in fr[:2]:
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    g.update_esdf(); g.update_esdf()
    o.update_esdf(); o.update_esdf()
    ig, ag = g.esdf_slice_image(1000.0); io, ao = o.esdf_slice_image(1000.0)
    assert ig.shape == io.shape and ig.size > 0 and np.array_equal(ag, ao) and np.abs(ig - io).max() <= TOL
    # and once more with new data between a pair of double updates
    for d, rgb, T in fr[2:]:
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
        g.update_esdf(); g.update_esdf(); o.update_esdf(); o.update_esdf()
    n, _ = compare_layer(M, g, o, M.LAYER_ESDF, oracle_mod.L_ESDF, fields_exact=ESDF_FIELDS)
    assert n > 10


def test_decay_excludes_the_camera_view_not_the_lidar_view(oracle_mod, hip_lib):
    """decayTsdfExcludeLastView<Camera> (nvblox_node.cpp:931-936): a LiDAR scan after the camera frame must not take the camera view's place."""
    from isaac_ros_nvblox_amd import mapper as M
    lidar = (128, 16, 0.1, -np.deg2rad(20.0), np.deg2rad(20.0))
    kw = dict(lidar_max_integration_distance_m=6.0, tsdf_decay_factor=0.5)
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 14); o = oracle_mod.OracleMap(po)
    sc = S.Scene()
    d, _, T = H.frames(1, H.SMALL_CAM, color=False)[0]
    g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    cam_view = H.idx_set(g.last_view())
    # a "LiDAR" in the same room, elsewhere, looking around
    Tl = np.eye(4, dtype=np.float32); Tl[:3, 3] = (-1.0, 0.5, 1.0)
    dirs = S.lidar_beam_dirs(lidar)
    rng_img = sc.raycast(Tl[:3, 3].astype(float), dirs.reshape(-1, 3)).reshape(dirs.shape[:2]).astype(np.float32)
    g.integrate_lidar_depth(rng_img, Tl, lidar); o.integrate_lidar_depth(rng_img, Tl, lidar)
    g.decay_tsdf(exclude_last_view=True); o.decay_tsdf(True)
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    # camera-only blocks kept weight 1; blocks only the LiDAR touched were decayed to 0.5
    lidar_view = H.idx_set(g.last_view())
    only_cam = sorted(cam_view - lidar_view); only_lidar = sorted(lidar_view - cam_view)
    assert only_cam and only_lidar
    bc, _ = g.get_blocks(M.LAYER_TSDF, only_cam); bl, _ = g.get_blocks(M.LAYER_TSDF, only_lidar)
    assert set(np.unique(bc["weight"]).tolist()) <= {0.0, 1.0}
    assert set(np.unique(bl["weight"]).tolist()) <= {0.0, 0.5}
    # blocks in BOTH views (the LiDAR scan re-claimed their shared view stamp after the camera frame): still the camera's
    # view, so not decayed -- weights are sums of the two undecayed measurements, never a halved one
    both = sorted(cam_view & lidar_view)
    assert len(both) > 20
    bb, _ = g.get_blocks(M.LAYER_TSDF, both)
    wb = set(np.unique(bb["weight"]).tolist())
    assert wb <= {0.0, 1.0, 2.0} and 2.0 in wb, wb
    for idx in both[:: max(1, len(both) // 40)]:
        assert set(np.unique(o.get_block(oracle_mod.L_TSDF, idx)["weight"]).tolist()) <= {0.0, 1.0, 2.0}, idx


def test_cleared_blocks_are_reported_for_decay_and_radius_clearing(oracle_mod, hip_lib):
    """Mapper::getClearedBlocks (layer_publishing.cpp:716,804): blocks deallocated by decay as well as by clearOutsideRadius."""
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.02, tsdf_decayed_weight_threshold=0.03)
    fr = H.frames(3, H.SMALL_CAM, color=False, stride=25)
    for d, rgb, T in fr:
        g.integrate_depth(d, T, H.SMALL_CAM); o.integrate_depth(d, T, H.SMALL_CAM)
    assert len(g.take_cleared_blocks()) == 0
    before = H.idx_set(g.block_indices(M.LAYER_TSDF))
    g.decay_tsdf(exclude_last_view=True); o.decay_tsdf(True)       # weight 1..3 * 0.02 < 0.03 for single-view blocks: deallocated
    after = H.idx_set(g.block_indices(M.LAYER_TSDF))
    cg = g.take_cleared_blocks(); co = o.take_cleared_blocks()
    assert len(cg) > 10 and np.array_equal(cg, co)
    assert H.idx_set(cg) == before - after
    assert len(g.take_cleared_blocks()) == 0                      # taken once
    g.clear_outside_radius((0.0, 0.0, 1.0), 1.5); o.clear_outside_radius((0.0, 0.0, 1.0), 1.5)
    after2 = H.idx_set(g.block_indices(M.LAYER_TSDF))
    cg = g.take_cleared_blocks(); co = o.take_cleared_blocks()
    assert len(cg) > 0 and np.array_equal(cg, co) and H.idx_set(cg) == after - after2


def test_load_map_rejects_a_corrupt_block_count_and_keeps_the_map(oracle_mod, hip_lib, tmp_path):
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 12)
    d, _, T = H.frames(1, H.SMALL_CAM, color=False)[0]
    g.integrate_depth(d, T, H.SMALL_CAM)
    n0 = g.num_blocks(M.LAYER_TSDF)
    p = tmp_path / "m.nvbx"
    g.save_map(p)
    raw = bytearray(p.read_bytes())
    # MapFileHeader = 8 + 4 + 4 + 4 + 4 bytes; first MapLayerHeader {u32 layer, u32 voxel_bytes, u64 n_blocks} follows
    struct.pack_into("<Q", raw, 24 + 8, (1 << 63) + 5)
    bad = tmp_path / "bad.nvbx"; bad.write_bytes(bytes(raw))
    with pytest.raises(M.NvbxError):
        g.load_map(bad)
    assert g.num_blocks(M.LAYER_TSDF) == n0                       # validation failed before the map was touched
    struct.pack_into("<I", raw, 16, 1 << 30)                      # implausible layer count
    bad.write_bytes(bytes(raw))
    with pytest.raises(M.NvbxError):
        g.load_map(bad)
    assert g.num_blocks(M.LAYER_TSDF) == n0


def test_set_blocks_with_duplicate_indices(hip_lib):
    """A batch that names the same block twice: both workgroups must find the slot (one of the two payloads wins)."""
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 10)
    idx = np.array([[1, 2, 3]] * 64 + [[4, 5, 6]], np.int32)
    data = np.zeros((65, 512), M.TSDF_DT)
    data["distance"] = 0.125; data["weight"] = 1.0
    for rep in range(20):
        g.clear()
        g.set_blocks(M.LAYER_TSDF, idx, data)
        assert g.num_blocks(M.LAYER_TSDF) == 2
        b, found = g.get_blocks(M.LAYER_TSDF, [[1, 2, 3], [4, 5, 6]])
        assert found.all() and (b["distance"] == np.float32(0.125)).all() and (b["weight"] == 1.0).all()


def test_nvblx_layer_cake_is_an_sqlite_database(oracle_mod, hip_lib, tmp_path):
    """saveLayerCake / loadMap on a *.nvblx path (nvblox_node.cpp:1663-1703): an SQLite file that Python's own sqlite3 module reads
    (an independent reader: block indices and voxel blobs equal the layer accessors'), and that loads back into an identical map."""
    import sqlite3
    M, g, o = make_pair(oracle_mod)
    for d, rgb, T in H.frames(3, H.SMALL_CAM, color=True, stride=9):
        g.integrate_depth(d, T, H.SMALL_CAM); g.integrate_color(rgb, T, H.SMALL_CAM)
    g.update_esdf()
    path = tmp_path / "map.nvblx"
    g.save_map(path)
    assert path.read_bytes()[:15] == b"SQLite format 3"
    db = sqlite3.connect(str(path))
    layers = {r[0]: r[1:] for r in db.execute("SELECT layer_type, voxel_size, block_size, voxel_bytes, num_blocks FROM layers")}
    assert set(layers) == {"tsdf_layer", "color_layer", "esdf_layer"}
    assert abs(layers["tsdf_layer"][0] - 0.05) < 1e-7 and abs(layers["tsdf_layer"][1] - 0.4) < 1e-6
    for name, layer, dt in (("tsdf_layer", M.LAYER_TSDF, M.TSDF_DT), ("color_layer", M.LAYER_COLOR, M.COLOR_DT), ("esdf_layer", M.LAYER_ESDF, M.ESDF_DT)):
        rows = db.execute("SELECT index_x, index_y, index_z, data FROM %s_blocks ORDER BY index_x, index_y, index_z" % name).fetchall()
        idx = g.block_indices(layer)
        assert layers[name][2] == dt.itemsize and layers[name][3] == len(idx) == len(rows)
        assert np.array_equal(np.array([r[:3] for r in rows], np.int32), idx)
        blocks, _ = g.get_blocks(layer, idx)
        for k in range(0, len(rows), 37):
            assert rows[k][3] == blocks[k].tobytes()
    db.close()
    g2 = M.Mapper(M.default_params(), block_capacity=1 << 12)
    g2.load_map(path)
    for layer, fields in ((M.LAYER_TSDF, ("distance", "weight")), (M.LAYER_COLOR, ("r", "g", "b", "weight")), (M.LAYER_ESDF, ESDF_FIELDS)):
        ia, ib = g.block_indices(layer), g2.block_indices(layer)
        assert np.array_equal(ia, ib)
        ba, _ = g.get_blocks(layer, ia); bb, _ = g2.get_blocks(layer, ib)
        for f in fields:
            assert np.array_equal(ba[f], bb[f]), (layer, f)
    # the loaded map keeps working: a further ESDF update gives the slice of the original
    g.update_esdf(); g2.update_esdf()
    sa, _ = g.esdf_slice_image(); sb, _ = g2.esdf_slice_image()
    assert sa.shape == sb.shape and np.array_equal(sa, sb)
    # a truncated / foreign SQLite file is refused and the map stays
    bad = tmp_path / "bad.nvblx"
    con = sqlite3.connect(str(bad)); con.execute("CREATE TABLE t(x)"); con.commit(); con.close()
    n0 = g2.num_blocks(M.LAYER_TSDF)
    with pytest.raises(M.NvbxError):
        g2.load_map(bad)
    assert g2.num_blocks(M.LAYER_TSDF) == n0
    # any other extension: the compact container, still loadable
    p2 = tmp_path / "map.bin"
    g.save_map(p2)
    assert p2.read_bytes()[:8] == b"NVBXMAP1"
    g2.load_map(p2)
    assert np.array_equal(g2.block_indices(M.LAYER_TSDF), g.block_indices(M.LAYER_TSDF))
