"""nvbx_integrate_depth_batch / nvbx_integrate_color_batch: n camera frames through ONE launch set.

The batch is DEFINED as equal to the n separate integrateDepth / integrateColor calls in order (include/nvblox_hip.h) -- the
reference's own multi-camera mode feeds up to four cameras through one mapper, one call per frame
(nvblox_ros/include/nvblox_ros/nvblox_node.hpp:298-332).  So: batch (HIP) == sequential (HIP) bit for bit, and == the oracle fed
sequentially within the usual tolerances; "last view" queries report the last camera's view."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_gpu_parity import TOL, compare_layer, make_pair

pytestmark = pytest.mark.gpu

ESDF_FIELDS = ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site")


def rig_frames(n_cams, k, cam=H.SMALL_CAM, noise=0.0):
    """Frame k of n cameras on the same rig circle at 45 degree yaw offsets (BASELINE.json configs[3]): heavily overlapping views."""
    sc = S.Scene(); rng = np.random.default_rng(100 + k)
    out = []
    for c in range(n_cams):
        T = S.trajectory_pose(k * 9, 200, yaw_offset_deg=45.0 * c)
        d, rgb = S.render(sc, T, cam, color=True, noise_sigma=noise, rng=rng)
        out.append((d, rgb, T))
    return out


def layers_equal(M, a, b, layer, fields):
    ia, ib = a.block_indices(layer), b.block_indices(layer)
    assert np.array_equal(ia, ib)
    ba, _ = a.get_blocks(layer, ia); bb, _ = b.get_blocks(layer, ib)
    for f in fields:
        assert np.array_equal(ba[f], bb[f]), (layer, f)
    return len(ia)


@pytest.mark.parametrize("n_cams", [2, 4, 8])
@pytest.mark.parametrize("kw", [dict(), dict(weighting_mode=4, invalid_depth_decay_factor=0.8)])
def test_batch_equals_separate_calls(oracle_mod, hip_lib, n_cams, kw):
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(**kw)
    gb = M.Mapper(pg, block_capacity=1 << 14)            # batched
    gs = M.Mapper(pg, block_capacity=1 << 14)            # separate calls
    o = oracle_mod.OracleMap(H.copy_params(pg, oracle_mod.OrcParams))
    for k in range(3):
        fr = rig_frames(n_cams, k, noise=0.004 if kw else 0.0)
        if kw:
            for d, _, _ in fr:
                d[20:40, 30:70] = 0.0                      # invalid depth: the decay path takes part in the per-voxel chain
        gb.integrate_depth_batch([d for d, _, _ in fr], [T for _, _, T in fr], cam)
        for d, _, T in fr:
            gs.integrate_depth(d, T, cam); o.integrate_depth(d, T, cam)
        assert H.idx_set(gb.last_view()) == H.idx_set(gs.last_view()) == H.idx_set(o.last_view())       # the LAST camera's view
        gb.integrate_color_batch([c for _, c, _ in fr], [T for _, _, T in fr], cam)
        for _, c, T in fr:
            gs.integrate_color(c, T, cam); o.integrate_color(c, T, cam)
        assert H.idx_set(gb.last_color_view()) == H.idx_set(gs.last_color_view()) == H.idx_set(o.last_color_view())
        assert np.array_equal(gb.synthetic_depth(), gs.synthetic_depth())
        gb.update_esdf(); gs.update_esdf(); o.update_esdf()
    n = layers_equal(M, gb, gs, M.LAYER_TSDF, ("distance", "weight"))
    assert n > 150
    layers_equal(M, gb, gs, M.LAYER_COLOR, ("r", "g", "b", "weight"))
    layers_equal(M, gb, gs, M.LAYER_ESDF, ESDF_FIELDS)
    compare_layer(M, gb, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    compare_layer(M, gb, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    compare_layer(M, gb, o, M.LAYER_ESDF, oracle_mod.L_ESDF, fields_exact=ESDF_FIELDS)
    # overlapping views really were fused: some voxels carry more than one update per step
    b, _ = gb.get_blocks(M.LAYER_TSDF, gb.block_indices(M.LAYER_TSDF))
    if not kw:
        assert (b["weight"] > 3.5).sum() > 1000
    # mesh of the batched map == mesh of the sequential map
    gb.update_color_mesh(); gs.update_color_mesh()
    ma, mb = gb.mesh(), gs.mesh()
    assert set(ma) == set(mb)
    for key in ma:
        assert np.array_equal(ma[key]["triangles"], mb[key]["triangles"]) and np.array_equal(ma[key]["vertices"], mb[key]["vertices"])
    assert gb.counters()["capacity_overflow"] == 0


def test_decay_after_a_batch_spares_the_last_cameras_view(oracle_mod, hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(tsdf_decay_factor=0.5)
    gb = M.Mapper(pg, block_capacity=1 << 14); gs = M.Mapper(pg, block_capacity=1 << 14)
    o = oracle_mod.OracleMap(H.copy_params(pg, oracle_mod.OrcParams))
    fr = rig_frames(4, 0)
    gb.integrate_depth_batch([d for d, _, _ in fr], [T for _, _, T in fr], cam)
    for d, _, T in fr:
        gs.integrate_depth(d, T, cam); o.integrate_depth(d, T, cam)
    gb.decay_tsdf(True); gs.decay_tsdf(True); o.decay_tsdf(True)
    layers_equal(M, gb, gs, M.LAYER_TSDF, ("distance", "weight"))
    compare_layer(M, gb, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    b, _ = gb.get_blocks(M.LAYER_TSDF, gb.block_indices(M.LAYER_TSDF))
    w = np.unique(b["weight"])
    assert 0.5 in w.tolist() and (w >= 1.0).any()           # blocks outside camera 3's view decayed, the spared ones did not


def test_batch_of_one_and_fallbacks(oracle_mod, hip_lib):
    """n = 1 is the plain call; a mapper with a freespace layer (per-frame time stamps) falls back to separate calls."""
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    fr = rig_frames(3, 1)
    for kw in (dict(), dict(projective_layer_type=2), dict(do_depth_preprocessing=1, depth_preprocessing_num_dilations=2)):
        pg = M.default_params(**kw)
        gb = M.Mapper(pg, block_capacity=1 << 14); gs = M.Mapper(pg, block_capacity=1 << 14)
        nb = 1 if not kw else 3
        gb.integrate_depth_batch([d for d, _, _ in fr[:nb]], [T for _, _, T in fr[:nb]], cam)
        for d, _, T in fr[:nb]:
            gs.integrate_depth(d, T, cam)
        assert layers_equal(M, gb, gs, M.LAYER_TSDF, ("distance", "weight")) > 50


def test_batch_argument_checks(hip_lib):
    from isaac_ros_nvblox_amd import mapper as M
    g = M.Mapper(M.default_params(), block_capacity=1 << 12)
    fr = rig_frames(2, 0)
    with pytest.raises(M.NvbxError):
        g.integrate_depth_batch([fr[0][0]] * 9, [fr[0][2]] * 9, H.SMALL_CAM)               # more than NVBX_MAX_BATCH
    with pytest.raises(M.NvbxError):
        g.integrate_depth_batch([fr[0][0], fr[1][0]], [fr[0][2], fr[1][2]], [H.SMALL_CAM, (80.0, 80.0, 79.5, 59.5, 161, 120)])   # camera != image
    with pytest.raises(M.NvbxError):                                                       # image sides above 32768 are refused (31-bit pixel indices)
        g.integrate_depth(np.ones((40000, 4), np.float32), fr[0][2], (80.0, 80.0, 1.5, 19999.5, 4, 40000))
    assert g.num_blocks(M.LAYER_TSDF) == 0


def test_batches_and_measurement_exchange_across_pool_growth(oracle_mod, hip_lib):
    """Small initial pools: the batch and the measurement-exchange paths grow them like the plain calls do, with the same result."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params()
    gb = M.Mapper(pg, block_capacity=1024); gm = M.Mapper(pg, block_capacity=1024); gs = M.Mapper(pg, block_capacity=1 << 14)
    all_buf = torch.zeros((4, 1024, M.Mapper.MEAS_BLOCK_BYTES), dtype=torch.uint8, device="cuda:0")
    all_cnt = torch.zeros((4,), dtype=torch.int32, device="cuda:0")
    for k in range(4):
        fr = rig_frames(4, k * 2)
        gb.integrate_depth_batch([d for d, _, _ in fr], [T for _, _, T in fr], cam); gb.synchronize()
        for d, _, T in fr:
            gs.integrate_depth(d, T, cam)
        # one mapper playing all four ranks: measure each camera, apply all
        tmp = [M.Mapper(pg, block_capacity=1024) for _ in range(4)]
        for r, (d, _, T) in enumerate(fr):
            tmp[r].measure_depth(d, T, cam, all_buf[r], all_cnt[r:r + 1]); tmp[r].synchronize()
        gm.apply_measurements(all_buf, all_cnt); gm.synchronize()
    assert gb.capacity > 1024 and gm.capacity > 1024
    assert gb.counters()["capacity_overflow"] == 0 and gm.counters()["capacity_overflow"] == 0
    n = layers_equal(M, gb, gs, M.LAYER_TSDF, ("distance", "weight"))
    layers_equal(M, gm, gs, M.LAYER_TSDF, ("distance", "weight"))
    assert n > 1100
