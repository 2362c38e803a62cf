"""The three decay switches the node can set (mapper_initialization.cpp:383-428: decay_integrator_deallocate_decayed_blocks,
tsdf_set_free_distance_on_decayed + tsdf_decayed_free_distance_vox, occupancy_decay_to_free).  [U] semantics; CPU: the
oracle's restatement does what the parameter names say; GPU: the HIP path against the oracle in every position, through ESDF and mesh."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

CAM = H.SMALL_CAM


def _fill(o, n=3, **kw):
    for d, rgb, T in H.frames(n, CAM, color=False, stride=20):
        o.integrate_depth(d, T, CAM)


def test_oracle_tsdf_set_free_distance_on_decayed(oracle_mod):
    kw = dict(tsdf_decay_factor=0.1, tsdf_decayed_weight_threshold=0.15, max_integration_distance_m=4.0)
    plain = oracle_mod.OracleMap(oracle_mod.default_params(**kw))
    free = oracle_mod.OracleMap(oracle_mod.default_params(tsdf_set_free_distance_on_decayed=1, tsdf_decayed_free_distance_vox=3.0, **kw))
    _fill(plain); _fill(free)
    before = {tuple(i): free.get_block(oracle_mod.L_TSDF, i).copy() for i in free.block_indices(oracle_mod.L_TSDF)}
    plain.decay_tsdf(False); free.decay_tsdf(False)          # weights 1..3 -> 0.1..0.3: voxels seen once fall below 0.15
    ip = H.idx_set(plain.block_indices(oracle_mod.L_TSDF)); iff = H.idx_set(free.block_indices(oracle_mod.L_TSDF))
    assert iff == ip and 0 < len(ip) < len(before)          # fully decayed blocks are deallocated either way
    n_free = 0
    for idx in sorted(ip):
        a = plain.get_block(oracle_mod.L_TSDF, idx); b = free.get_block(oracle_mod.L_TSDF, idx); w0 = before[idx]["weight"]
        dec = (w0 > 0) & (w0 * np.float32(0.1) < np.float32(0.15))          # observed, now below the threshold
        assert np.array_equal(b["weight"][~dec], a["weight"][~dec]) and np.array_equal(b["distance"][~dec], a["distance"][~dec])
        assert np.all(b["weight"][dec] == np.float32(0.15)) and np.all(b["distance"][dec] == np.float32(3.0) * np.float32(0.05))
        assert np.all(b["weight"][w0 == 0] == 0)               # never-observed voxels stay unknown
        n_free += int(dec.sum())
    assert n_free > 1000


def test_oracle_keep_decayed_blocks_and_occupancy_decay_to_free(oracle_mod):
    kw = dict(tsdf_decay_factor=0.01, tsdf_decayed_weight_threshold=0.05, max_integration_distance_m=4.0)
    keep = oracle_mod.OracleMap(oracle_mod.default_params(decay_deallocate_decayed_blocks=0, **kw))
    drop = oracle_mod.OracleMap(oracle_mod.default_params(**kw))
    _fill(keep); _fill(drop)
    n0 = keep.num_blocks()
    keep.decay_tsdf(False); drop.decay_tsdf(False)
    assert keep.num_blocks() == n0 and drop.num_blocks() == 0 and len(keep.take_cleared_blocks()) == 0 and len(drop.take_cleared_blocks()) == n0
    w = np.concatenate([keep.get_block(oracle_mod.L_TSDF, i)["weight"] for i in keep.block_indices(oracle_mod.L_TSDF)])
    assert w.max() < 0.05 and (w > 0).sum() > 1000
    # occupancy: decay to free
    occ = dict(projective_layer_type=1, free_region_occupancy_probability=0.3, occupied_region_occupancy_probability=0.9, max_integration_distance_m=4.0,
               free_region_decay_probability=0.6, occupied_region_decay_probability=0.2)
    a = oracle_mod.OracleMap(oracle_mod.default_params(**occ)); b = oracle_mod.OracleMap(oracle_mod.default_params(occupancy_decay_to_free=1, **occ))
    _fill(a, 2); _fill(b, 2)
    before = {tuple(i): b.get_block(oracle_mod.L_TSDF, i)["distance"].copy() for i in b.block_indices(oracle_mod.L_TSDF)}
    for _ in range(6):
        a.decay_occupancy(); b.decay_occupancy()
    ib = H.idx_set(b.block_indices(oracle_mod.L_TSDF))
    known = {idx for idx, v0 in before.items() if (v0 != 0).any()}        # (all-unknown blocks are deallocated by the first pass in either mode)
    assert ib == known and len(H.idx_set(a.block_indices(oracle_mod.L_TSDF))) < len(ib)      # to-free: nothing fades to unknown, no known block dies
    n_occ = 0
    for idx, v0 in before.items():
        if idx not in known:
            continue
        v = b.get_block(oracle_mod.L_TSDF, np.array(idx, np.int32))["distance"]
        assert np.array_equal(v[v0 < 0], v0[v0 < 0])                     # free voxels untouched
        assert np.all(v[v0 > 0] < 0) and np.all(v[v0 == 0] == 0)         # occupied voxels ended up free; unknown stays unknown
        n_occ += int((v0 > 0).sum())
    assert n_occ > 200


SWITCHES = [dict(tsdf_set_free_distance_on_decayed=1, tsdf_decayed_free_distance_vox=3.0),
            dict(decay_deallocate_decayed_blocks=0),
            dict(decay_deallocate_decayed_blocks=0, tsdf_set_free_distance_on_decayed=1)]


@pytest.mark.gpu
@pytest.mark.parametrize("sw", SWITCHES, ids=["set_free", "keep_blocks", "keep_blocks+set_free"])
def test_tsdf_decay_switches_parity(oracle_mod, hip_lib, sw):
    from test_gpu_parity import make_pair, compare_layer, TOL
    from test_gpu_sequences import check_all
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.3, tsdf_decayed_weight_threshold=0.2, max_integration_distance_m=5.0, invalid_depth_decay_factor=0.8, **sw)
    sc = S.Scene()
    fr = H.frames(8, CAM, stride=11)
    for k, (d, rgb, T) in enumerate(fr):
        g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
        g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
        if k % 2 == 1:
            ex = bool(k % 4 == 1)
            g.decay_tsdf(ex); o.decay_tsdf(ex)
            g.decay_tsdf(False); o.decay_tsdf(False)
            assert np.array_equal(g.take_cleared_blocks(), o.take_cleared_blocks())
        if k % 3 == 2:
            g.update_esdf(); o.update_esdf()
    g.update_esdf(); o.update_esdf(); g.update_color_mesh(full=True); o.update_mesh(full=True)
    check_all(M, oracle_mod, g, o)
    idx = g.block_indices(M.LAYER_TSDF); b, _ = g.get_blocks(M.LAYER_TSDF, idx)
    if sw.get("tsdf_set_free_distance_on_decayed"):
        fd = np.float32(sw.get("tsdf_decayed_free_distance_vox", 4.0)) * np.float32(0.05)
        assert ((b["distance"] == fd) & (b["weight"] == np.float32(0.2))).sum() > 500          # voxels that decayed to free
    if not sw.get("decay_deallocate_decayed_blocks", 1):
        assert len(g.take_cleared_blocks()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("sw", [dict(occupancy_decay_to_free=1), dict(decay_deallocate_decayed_blocks=0), dict(occupancy_decay_to_free=1, decay_deallocate_decayed_blocks=0)],
                         ids=["to_free", "keep_blocks", "to_free+keep_blocks"])
def test_occupancy_decay_switches_parity(oracle_mod, hip_lib, sw):
    from test_gpu_parity import make_pair, compare_occupancy, TOL
    occ = dict(projective_layer_type=1, free_region_occupancy_probability=0.3, occupied_region_occupancy_probability=0.9, max_integration_distance_m=5.0,
               free_region_decay_probability=0.6, occupied_region_decay_probability=0.2)
    M, g, o = make_pair(oracle_mod, **occ, **sw)
    for k, (d, rgb, T) in enumerate(H.frames(6, CAM, color=False, stride=13)):
        g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
        for _ in range(3):
            g.decay_occupancy(); o.decay_occupancy()
        if k % 2:
            g.update_esdf(); o.update_esdf()
    n, bg = compare_occupancy(M, g, o, oracle_mod)
    assert n > 50
    g.update_esdf(); o.update_esdf()
    sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
