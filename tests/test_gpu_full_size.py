"""Parity at the REAL size of the two BASELINE.json configurations the other files only reach scaled down:
configs[4] -- 1024 x 64 spinning LiDAR, 0.10 m voxels, 200 m range (~112 k blocks per scan: hash load, pool growth, LDS-set
spill and early-flush paths of k_mark_view<Lidar>, nvblox_node.cpp:1382-1384) and configs[2] -- the dynamic / decay frame at
640 x 480 (lock-free connected components over 307 k pixels, freespace over a room-sized view, nvblox_node.cpp:1062 with
mapping_type dynamic, mapper_initialization.cpp:27-109).  The HIP path against the oracle, bit for bit."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_lidar_full_config_parity_against_oracle(oracle_mod, hip_lib):
    """Two scans of configs[4] from two poses.  View sets equal on all ~112 k blocks per scan, every voxel of EVERY block equal
    (bit for bit: distance and weight), and the compared set is checked to contain blocks beyond 150 m and blocks on the
    azimuth seam (the -x axis of the sensor, where image column 0 meets column 1023)."""
    from isaac_ros_nvblox_amd import mapper as M
    kw = dict(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2, weighting_mode=0)
    pg = M.default_params(**kw); po = H.copy_params(pg, oracle_mod.OrcParams)
    g = M.Mapper(pg, block_capacity=1 << 17)                   # grows once on the way (146 k blocks after two scans)
    o = oracle_mod.OracleMap(po)
    sc = S.LidarScene()
    poses = [S.lidar_pose(0), S.lidar_pose(3)]
    for T in poses:
        img = S.render_lidar(sc, T, S.SPINNING_LIDAR, max_range=200.0)
        g.integrate_lidar_depth(img, T, S.SPINNING_LIDAR); o.integrate_lidar_depth(img, T, S.SPINNING_LIDAR)
        vg = np.asarray(g.last_view()).reshape(-1, 3); vo = np.asarray(o.last_view()).reshape(-1, 3)
        assert len(vo) > 100000
        assert len(vg) == len(vo) and H.idx_set(vg) == H.idx_set(vo)
    assert g.counters()["capacity_overflow"] == 0
    ig = g.block_indices(M.LAYER_TSDF); io = o.block_indices(oracle_mod.L_TSDF)
    assert np.array_equal(ig, io) and len(io) > 112000
    # which blocks are far / on the seam (of the last scan's sensor frame)
    T = np.asarray(poses[-1], np.float64)
    ctr = (io.astype(np.float64) + 0.5) * 0.8
    ps = (ctr - T[:3, 3]) @ T[:3, :3]                           # block centres in the sensor frame
    far = np.hypot(ps[:, 0], ps[:, 1]) > 150.0
    seam = (ps[:, 0] < -5.0) & (np.abs(ps[:, 1]) < 0.8)
    assert far.sum() > 2000 and seam.sum() > 100
    n_obs = 0; n_far_obs = 0; n_seam_obs = 0
    CH = 8192
    for s0 in range(0, len(io), CH):
        sel = io[s0:s0 + CH]
        bg, found = g.get_blocks(M.LAYER_TSDF, sel)
        assert found.all()
        for k, idx in enumerate(sel):
            b = o.get_block(oracle_mod.L_TSDF, idx)
            if not (np.array_equal(bg[k]["distance"], b["distance"]) and np.array_equal(bg[k]["weight"], b["weight"])):
                bad = np.nonzero((bg[k]["distance"] != b["distance"]) | (bg[k]["weight"] != b["weight"]))[0]
                raise AssertionError("block %s: %d voxels differ, first %d: gpu (%r, %r) oracle (%r, %r)" % (
                    idx, len(bad), bad[0], bg[k]["distance"][bad[0]], bg[k]["weight"][bad[0]], b["distance"][bad[0]], b["weight"][bad[0]]))
            nz = int((b["weight"] > 0).sum())
            n_obs += nz
            if far[s0 + k]: n_far_obs += nz
            if seam[s0 + k]: n_seam_obs += nz
    assert n_obs > 5_000_000 and n_far_obs > 10_000 and n_seam_obs > 5_000, (n_obs, n_far_obs, n_seam_obs)


def _scipy_filter(mk, thr):
    import scipy.ndimage as ndi
    lab, n = ndi.label(mk, structure=np.ones((3, 3)))
    sizes = ndi.sum(mk, lab, index=np.arange(1, n + 1))
    want = mk.copy(); want[np.isin(lab, np.nonzero(sizes < thr)[0] + 1)] = 0
    return want


@pytest.mark.parametrize("density", [0.2, 0.42, 0.55])
def test_remove_small_components_full_res_vs_oracle_and_scipy(oracle_mod, hip_lib, density):
    """removeSmallConnectedComponents at 480 x 640 (the lock-free union-find of dynamics.hip: ~10^5 concurrent unions, long
    snaking components at the percolation threshold 0.42) at four thresholds against the oracle AND scipy.ndimage.label."""
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 12)
    rng = np.random.default_rng(int(density * 100))
    for rep in range(3):
        mk = (rng.random((480, 640)) < density).astype(np.uint8)
        if rep == 1:                                            # large solid regions + a frame-spanning ring
            mk[100:300, 200:500] = 1; mk[0, :] = 1; mk[-1, :] = 1; mk[:, 0] = 1; mk[:, -1] = 1
        if rep == 2:                                            # a one-pixel serpentine through the whole image (one huge, thin component)
            mk[:] = 0
            for r in range(0, 480, 4):
                mk[r, :] = 1; mk[r + 1:r + 4, (639 if (r // 4) % 2 == 0 else 0)] = 1
            mk[rng.random(mk.shape) < 0.02] = 1
        for thr in (2, 9, 60, 400):
            got = g.remove_small_components(mk, thr).cpu().numpy()
            assert np.array_equal(got, oracle_mod.remove_small_components(mk, thr)), (density, rep, thr)
            assert np.array_equal(got, _scipy_filter(mk, thr)), (density, rep, thr)
