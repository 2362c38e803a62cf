"""Sequence-level parity: the HIP path against the oracle over (i) randomised interleavings of every entry point of the
hot path -- integrateDepth / integrateColor / updateEsdf / updateColorMesh / decay / clearOutsideRadius /
clearTsdfInsideShapes -- from random poses, and (ii) the Redwood-like configuration of BASELINE.json configs[2]
(SURVEY.md 8d): a moving box, tsdf decay every 6th frame, invalid-depth decay 0.8."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S
from test_gpu_parity import make_pair, compare_layer, TOL

pytestmark = pytest.mark.gpu
CAM = H.SMALL_CAM


def check_all(M, oracle_mod, g, o, mesh=True):
    compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    assert H.idx_set(g.block_indices(M.LAYER_ESDF)) == H.idx_set(o.block_indices(oracle_mod.L_ESDF))
    if len(o.block_indices(oracle_mod.L_ESDF)):
        sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
        assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL and np.allclose(ag, ao, atol=1e-6)
    if mesh:
        mg = g.mesh()
        n_tri = 0
        for idx in o.block_indices(oracle_mod.L_TSDF):
            mo = o.mesh_block(idx)
            a = mg.get(tuple(idx))
            if a is None:
                assert mo is None or len(mo["triangles"]) == 0, idx
                continue
            assert np.array_equal(a["triangles"], mo["triangles"]), idx
            if len(mo["vertices"]):
                assert np.abs(a["vertices"] - mo["vertices"]).max() <= TOL
            n_tri += len(mo["triangles"])
        return n_tri
    return 0


@pytest.mark.parametrize("deferral", [False, True], ids=["classic", "colour-deferral"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_api_sequence_parity(oracle_mod, hip_lib, seed, deferral):
    """deferral = nvbx_mapper_set_color_deferral: integrateColor / updateEsdf held back and carried out by the next integrateDepth in
    pipelined order, or replayed by whatever other entry point comes first -- the same map either way"""
    rng = np.random.default_rng(100 + seed)
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.7, tsdf_decayed_weight_threshold=0.2,
                        invalid_depth_decay_factor=(0.8 if seed % 3 == 2 else -1.0), weighting_mode=(4 if seed % 3 == 1 else 0))
    g.set_color_deferral(deferral)
    sc = S.Scene()
    n_ops = {"depth": 0, "color": 0, "esdf": 0, "mesh": 0, "decay": 0, "radius": 0, "shapes": 0}
    last_T = None
    for step in range(36):
        op = rng.choice(["depth", "depth", "depth", "color", "color", "esdf", "esdf", "mesh", "decay", "radius", "shapes"])
        if step < 3:
            op = "depth"
        n_ops[op] += 1
        if op in ("depth", "color"):
            i = int(rng.integers(0, 200))
            T = S.trajectory_pose(i, 200, radius=float(rng.uniform(0.3, 1.6)), height=float(rng.uniform(0.8, 2.0)),
                                  pitch_deg=float(rng.uniform(-35.0, 15.0)), yaw_offset_deg=float(rng.uniform(-60.0, 60.0)))
            d, rgb = S.render(sc, T, CAM, max_range=(6.0 if rng.random() < 0.3 else None))
            if op == "depth":
                g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
                assert H.idx_set(g.last_view()) == H.idx_set(o.last_view())
                last_T = T
            else:
                g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
        elif op == "esdf":
            g.update_esdf(); o.update_esdf()
        elif op == "mesh":
            full = bool(rng.random() < 0.3)
            g.update_color_mesh(full=full); o.update_mesh(full=full)
        elif op == "decay":
            ex = bool(rng.random() < 0.5)
            g.decay_tsdf(ex); o.decay_tsdf(ex)
        elif op == "radius":
            c = (float(last_T[0, 3]), float(last_T[1, 3]), float(last_T[2, 3]))
            r = float(rng.uniform(2.0, 4.0))
            g.clear_outside_radius(c, r); o.clear_outside_radius(c, r)
        elif op == "shapes":
            ctr = tuple(float(v) for v in rng.uniform([-2.5, -2.0, 0.2], [2.5, 2.0, 2.0]))
            lo = tuple(float(v) for v in rng.uniform([-3.0, -2.5, 0.0], [2.0, 1.5, 1.0]))
            shapes = [("sphere", ctr, float(rng.uniform(0.3, 0.9))), ("aabb", lo, tuple(v + float(rng.uniform(0.3, 1.2)) for v in lo))]
            g.clear_tsdf_inside_shapes(shapes); o.clear_tsdf_inside_shapes(shapes)
        if step % 9 == 8:
            compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    g.update_esdf(); o.update_esdf(); g.update_color_mesh(full=True); o.update_mesh(full=True)   # (mesh() = the blocks of the last update)
    n_tri = check_all(M, oracle_mod, g, o)
    assert n_tri > 500 and g.counters()["capacity_overflow"] == 0
    assert n_ops["depth"] >= 5


@pytest.mark.parametrize("CAM", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_redwood_like_decay_dynamic_sequence(oracle_mod, hip_lib, CAM):
    """configs[2] (at the reduced and at BASELINE.json's real image size): 8 x 6 x 2.8 m room, a box translating at 0.5 m/s, decay 0.95 every 6th frame (5 Hz at 30 Hz input),
    invalid_depth_decay_factor 0.8, depth limited to 5 m so that part of every frame is invalid."""
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.95, invalid_depth_decay_factor=0.8, max_integration_distance_m=5.0)
    for i in range(30):
        sc = S.redwood_like_scene(i * 6)                       # every 6th frame of a 30 Hz stream: the box moves 0.1 m per step
        T = S.trajectory_pose(i * 5, 200, radius=1.2, height=1.4)
        d, rgb = S.render(sc, T, CAM, max_range=5.0)
        g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
        g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
        g.decay_tsdf(True); o.decay_tsdf(True)
        if i % 3 == 2:
            g.update_esdf(); o.update_esdf()
        if i % 10 == 9:
            g.update_color_mesh(); o.update_mesh()
    g.update_esdf(); o.update_esdf(); g.update_color_mesh(full=True); o.update_mesh(full=True)
    n_tri = check_all(M, oracle_mod, g, o)
    assert n_tri > 1000
    # the decay really acted: observed weights below the integer ladder of the constant weighting
    idx = g.block_indices(M.LAYER_TSDF)
    b, _ = g.get_blocks(M.LAYER_TSDF, idx)
    w = b["weight"][b["weight"] > 0]
    assert (np.abs(w - np.round(w)) > 1e-3).mean() > 0.5


@pytest.mark.parametrize("deferral", [False, True], ids=["classic", "colour-deferral"])
def test_long_soak_parity(oracle_mod, hip_lib, deferral):
    """400 frames of a wandering camera with colour, an ESDF update every 3rd frame, decay every 7th, radius clearing every 40th and
    a mesh update every 25th -- thousands of block allocations, deallocations, slot re-use and hash rebuilds; the maps are
    compared with the oracle every 50 frames.  Guards the device-side allocator / hash / work lists against rare races."""
    rng = np.random.default_rng(7)
    M, g, o = make_pair(oracle_mod, tsdf_decay_factor=0.8, tsdf_decayed_weight_threshold=0.15, max_integration_distance_m=5.0)
    g.set_color_deferral(deferral)
    sc = S.Scene()
    pos = np.array([0.0, 0.0, 1.4]); yaw = 0.0
    live_hist = []
    for f in range(400 if not deferral else 160):          # (the pipelined variant: 160 frames, enough for slot re-use and hash rebuilds)
        yaw += float(rng.uniform(-0.25, 0.35)); pos[:2] += rng.uniform(-0.08, 0.08, 2); pos[:2] = np.clip(pos[:2], -1.8, 1.8)
        pos[2] = float(np.clip(pos[2] + rng.uniform(-0.03, 0.03), 0.9, 2.0))
        T = S.look_pose(pos.copy(), yaw, float(rng.uniform(-0.5, 0.2)))
        d, rgb = S.render(sc, T, CAM, max_range=(5.0 if f % 5 else None))
        g.integrate_depth(d, T, CAM); o.integrate_depth(d, T, CAM)
        g.integrate_color(rgb, T, CAM); o.integrate_color(rgb, T, CAM)
        if f % 3 == 2:
            g.update_esdf(); o.update_esdf()
        if f % 7 == 6:
            g.decay_tsdf(True); o.decay_tsdf(True)
        if f % 40 == 39:
            c = (float(pos[0]), float(pos[1]), float(pos[2]))
            g.clear_outside_radius(c, 3.0); o.clear_outside_radius(c, 3.0)
        if f % 25 == 24:
            g.update_color_mesh(); o.update_mesh()
        if f % 50 == 49:
            compare_layer(M, g, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
            compare_layer(M, g, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
            g.update_esdf(); o.update_esdf()
            sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
            assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL, f
            live_hist.append(g.num_blocks(M.LAYER_TSDF))
    g.update_color_mesh(full=True); o.update_mesh(full=True)
    assert check_all(M, oracle_mod, g, o) > 1000
    c = g.counters()
    assert c["capacity_overflow"] == 0 and min(live_hist) > 200 and c["blocks_allocated"] < (1 << 14)
