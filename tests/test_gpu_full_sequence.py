"""BASELINE.json configs[1] at FULL size on the GPU (640x480 @ 0.05 m, fuser.yaml parameters, the 200-frame circle of
SURVEY 8d), checked through size-independent properties because the scalar oracle needs ~20 ms per frame:
analytic scene ground truth for TSDF and mesh, a brute-force Euclidean distance transform of the GPU's own site set for
the ESDF, slice/occupancy consistency, and an oracle spot check on a strided subset of frames."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_oracle_ground_truth import scene_sdf

pytestmark = pytest.mark.gpu

CAM = S.REPLICA_LIKE_CAM


@pytest.fixture(scope="module")
def full_map(hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 15)
    sc = S.Scene()
    for i in range(0, 200, 4):                     # 50 frames spread over the whole loop
        T = S.trajectory_pose(i, 200)
        d, rgb = S.render(sc, T, CAM)
        g.integrate_depth(d, T, CAM); g.integrate_color(rgb, T, CAM)
        if i % 20 == 0:
            g.update_esdf()                        # incremental updates in between
    g.update_esdf(); g.update_color_mesh(full=True)
    return M, g


def test_full_size_tsdf_close_to_analytic_distance(full_map):
    M, g = full_map
    idx = g.block_indices(M.LAYER_TSDF)
    assert len(idx) > 1500 and g.counters()["capacity_overflow"] == 0
    b, found = g.get_blocks(M.LAYER_TSDF, idx)
    assert found.all()
    vs, trunc = 0.05, 0.2
    gx, gy, gz = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    off = np.stack([gx, gy, gz], -1).reshape(-1, 3)                      # [x][y][z] -> linear z + 8y + 64x
    p = (idx[:, None, :] * 8 + off[None, :, :] + 0.5) * vs
    sel = (b["weight"] >= 3.0) & (np.abs(b["distance"]) < 0.5 * trunc)
    err = np.abs(b["distance"][sel] - scene_sdf(p[sel]))
    assert err.size > 50000
    assert np.median(err) < 0.4 * vs and np.percentile(err, 99) < 2.5 * vs
    # weights are clamped, distances inside the truncation band
    assert b["weight"].max() <= 5.0 + 1e-6 and np.abs(b["distance"]).max() <= trunc + 1e-6


def test_full_size_esdf_is_exact_edt_of_its_sites(full_map):
    M, g = full_map
    idx = g.block_indices(M.LAYER_ESDF)
    assert len(idx) > 150
    b, _ = g.get_blocks(M.LAYER_ESDF, idx)
    bx0, by0 = idx[:, 0].min(), idx[:, 1].min()
    W, Hh = (idx[:, 0].max() - bx0 + 1) * 8, (idx[:, 1].max() - by0 + 1) * 8
    site = np.zeros((Hh, W), bool); alloc = np.zeros((Hh, W), bool); sq = np.zeros((Hh, W), np.float32)
    vz = 1                                                               # floor(0.09 / 0.05)
    for k, i in enumerate(idx):
        blk = b[k].reshape(8, 8, 8)[:, :, vz]
        ys, xs = (i[1] - by0) * 8, (i[0] - bx0) * 8
        site[ys:ys + 8, xs:xs + 8] = blk["is_site"].T.astype(bool)
        alloc[ys:ys + 8, xs:xs + 8] = True
        sq[ys:ys + 8, xs:xs + 8] = blk["squared_distance_vox"].T
    sy, sx = np.nonzero(site)
    assert len(sy) > 500
    yy, xx = np.nonzero(alloc)
    best = np.full(len(yy), np.inf)
    for s0 in range(0, len(sy), 2000):                                   # chunked brute force
        d2 = (yy[:, None] - sy[None, s0:s0 + 2000]) ** 2 + (xx[:, None] - sx[None, s0:s0 + 2000]) ** 2
        best = np.minimum(best, d2.min(axis=1))
    max_sq = np.float32((np.float32(2.0) / np.float32(0.05)) ** 2)
    want = np.where(best <= max_sq, best, max_sq).astype(np.float32)
    assert np.array_equal(sq[yy, xx], want)
    # the slice image is the same field in metres, unknown elsewhere; occupancy follows it
    img, aabb = g.esdf_slice_image_device(1000.0)
    host = img.cpu().numpy()
    assert host.shape == (Hh, W) and abs(aabb[0] - bx0 * 0.4) < 1e-6 and abs(aabb[1] - by0 * 0.4) < 1e-6
    known = np.abs(host - 1000.0) >= 1e-2
    assert np.allclose(np.abs(host[known]), np.sqrt(sq[known]) * np.float32(0.05), atol=1e-6)
    occ = g.occupancy_grid_from_slice(img, 1000.0).cpu().numpy()
    assert ((occ == -1) == ~known).all() and ((occ == 100) == (known & (host <= 0))).all()


def test_full_size_mesh_on_the_analytic_surface(full_map):
    M, g = full_map
    mesh = g.mesh()
    v = np.concatenate([m["vertices"] for m in mesh.values() if len(m["vertices"])])
    n = np.concatenate([m["normals"] for m in mesh.values() if len(m["vertices"])])
    t_total = sum(len(m["triangles"]) for m in mesh.values())
    assert len(v) > 30000 and t_total > 50000
    d = np.abs(scene_sdf(v.astype(np.float64)))
    assert np.median(d) < 0.01 and np.percentile(d, 99) < 0.05          # within a voxel of the true surfaces
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-3)
    for m in mesh.values():                                              # triangle indices are local and in range
        if len(m["triangles"]):
            assert m["triangles"].min() >= 0 and m["triangles"].max() < len(m["vertices"])
    # colours come from the procedural texture {64, 192} (or the neutral grey where no colour was integrated)
    c = np.concatenate([m["colors"] for m in mesh.values() if len(m["vertices"])])
    assert c[:, 3].min() == 255
    assert np.isin(c[:, :3], np.arange(60, 200)).mean() > 0.99


def test_full_size_oracle_spot_check(oracle_mod, hip_lib):
    """Strided subset of the same sequence against the oracle, full resolution."""
    from isaac_ros_nvblox_amd import mapper as M
    pg = M.default_params(); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 15); o = oracle_mod.OracleMap(po)
    sc = S.Scene()
    for i in (0, 50, 100, 150):
        T = S.trajectory_pose(i, 200)
        d, rgb = S.render(sc, T, CAM)
        g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
        g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
        assert np.abs(g.synthetic_depth() - o.synthetic_depth()).max() <= 1e-4
    g.update_esdf(); o.update_esdf()
    ig = g.block_indices(M.LAYER_TSDF); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io)
    bg, _ = g.get_blocks(M.LAYER_TSDF, ig)
    cg, cf = g.get_blocks(M.LAYER_COLOR, ig)
    for k, idx in enumerate(io):
        bo = o.get_block(oracle_mod.L_TSDF, idx)
        assert np.abs(bg[k]["distance"] - bo["distance"]).max() <= 1e-4 and np.abs(bg[k]["weight"] - bo["weight"]).max() <= 1e-4
        co = o.get_block(oracle_mod.L_COLOR, idx)
        assert (co is not None) == bool(cf[k])
        if co is not None:
            assert np.abs(cg[k]["r"].astype(int) - co["r"].astype(int)).max() <= 1 and np.abs(cg[k]["weight"] - co["weight"]).max() <= 1e-4
    sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= 1e-4
