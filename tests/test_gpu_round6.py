"""Round 6: nvbx_integrate_depth_pair -- the background and the foreground mapper of a MultiMapper's dynamic / human mapping types take the two halves of
one mask-split depth frame (nvblox_node.cpp:1057-1062) in TWO launches instead of four (k_mark_view_pair, k_integrate_tsdf_color_pair).  Defined as
equal to the two nvbx_integrate_depth calls in order: every layer of both maps bit for bit, whatever each mapper holds back at the time (colour frame +
ESDF update, ESDF update alone, nothing), across decay, clearing, queries and drains; and equal to the checker."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _layers(M, m, layers):
    out = {}
    for name, lay in layers:
        idx = m.block_indices(lay)
        blk, _ = m.get_blocks(lay, idx)
        out[name] = (idx, blk)
    return out


def _same(M, a, b, layers, tag):
    la, lb = _layers(M, a, layers), _layers(M, b, layers)
    for name in la:
        ia, ba = la[name]; ib, bb = lb[name]
        assert np.array_equal(ia, ib), (tag, name, len(ia), len(ib))
        assert ba.tobytes() == bb.tobytes(), (tag, name)


def _short(name):
    import bench
    return bench.short(name)


@pytest.mark.parametrize("cam", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_depth_pair_equals_the_two_calls_and_the_checker(oracle_mod, hip_lib, cam):
    """The dynamic-mapping frame as bench.py --workload decay drives it: front end, static mapper (TSDF + freespace, colour, ESDF), dynamic mapper
    (occupancy, ESDF), decay on every sixth frame, a slice query now and then -- once with nvbx_integrate_depth_pair, once with the two calls, once on
    the checker.  The held-back state of each mapper differs from frame to frame (colour on two frames of three, an ESDF update skipped now and then)."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    rows, cols = cam[5], cam[4]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    fs = dict(projective_layer_type=2, max_integration_distance_m=5.0, invalid_depth_decay_factor=0.8, tsdf_decay_factor=0.95, min_duration_since_occupied_for_freespace_ms=250)
    oc = dict(projective_layer_type=1, max_integration_distance_m=5.0, free_region_decay_probability=0.55, occupied_region_decay_probability=0.4)
    ps, pd = M.default_params(**fs), M.default_params(**oc)
    with torch.cuda.stream(stream):
        sa = M.Mapper(ps, block_capacity=1 << 14, stream=stream.cuda_stream); da = M.Mapper(pd, block_capacity=1 << 12, stream=stream.cuda_stream)      # the pair
        sb = M.Mapper(ps, block_capacity=1 << 14, stream=stream.cuda_stream); db = M.Mapper(pd, block_capacity=1 << 12, stream=stream.cuda_stream)      # the two calls
        os_ = oracle_mod.OracleMap(H.copy_params(ps, oracle_mod.OrcParams)); od = oracle_mod.OracleMap(H.copy_params(pd, oracle_mod.OrcParams))
        sa.set_profiling(True); da.set_profiling(True)
        eye = np.eye(4, dtype=np.float32)
        static_scene = S.Scene(); moving = S.Scene(box_min=(1.6, -0.3, 0.0), box_max=(2.0, 0.3, 1.3))
        min_component = 40 if cols == 160 else 640
        rng = np.random.default_rng(6)
        n_frames = 20
        for i in range(n_frames):
            sc = static_scene if i < 8 else moving
            T = S.trajectory_pose(min(i, 10), 200)
            d, rgb = S.render(sc, T, cam, max_range=5.0)
            d_dev = torch.from_numpy(d).to(dev); rgb_dev = torch.from_numpy(rgb).to(dev)
            t_ms = i * 100
            halves = []
            for st_ in (sa, sb):
                st_.set_time_ms(t_ms)
                mk = torch.empty((rows, cols), dtype=torch.uint8, device=dev); un = torch.empty((rows, cols), dtype=torch.float32, device=dev); ma = torch.empty_like(un)
                st_.dynamic_depth_split_into(d_dev, T, cam, 5.0, min_component, 0.25, mk, un, ma)
                halves.append((un, ma))
            sa.integrate_depth_pair(halves[0][0], da, halves[0][1], T, cam)
            sb.integrate_depth(halves[1][0], T, cam); db.integrate_depth(halves[1][1], T, cam)
            mo = os_.detect_dynamics(d, T, cam, 5.0); mo = oracle_mod.remove_small_components(mo, min_component)
            uo, mao = oracle_mod.split_depth_by_mask(d, mo, eye, cam, cam, 0.25)
            os_.set_time_ms(t_ms); os_.integrate_depth(uo, T, cam); od.integrate_depth(mao, T, cam)
            if i % 3 != 2:
                for st_ in (sa, sb):
                    st_.integrate_color(rgb_dev, T, cam)
                os_.integrate_color(rgb, T, cam)
            if rng.random() < 0.8:
                for m_ in (sa, sb, da, db):
                    m_.update_esdf()
                os_.update_esdf(); od.update_esdf()
            if i % 6 == 5:
                sa.decay_tsdf(True); da.decay_occupancy(); sb.decay_tsdf(True); db.decay_occupancy()
                os_.decay_tsdf(True); od.decay_occupancy()
            if i % 7 == 3:                                           # a query: drains what is held back
                img_a, _ = sa.esdf_slice_image(); img_b, _ = sb.esdf_slice_image()
                assert img_a.shape == img_b.shape and np.array_equal(img_a, img_b), i
        for m_ in (sa, sb, da, db):
            m_.synchronize()
        static_layers = [("tsdf", M.LAYER_TSDF), ("color", M.LAYER_COLOR), ("esdf", M.LAYER_ESDF), ("freespace", M.LAYER_FREESPACE)]
        dyn_layers = [("occupancy", M.LAYER_OCCUPANCY), ("esdf", M.LAYER_ESDF)]
        _same(M, sa, sb, static_layers, "static mapper, pair vs two calls")
        _same(M, da, db, dyn_layers, "dynamic mapper, pair vs two calls")
        # ... and the checker: block sets equal, TSDF / occupancy values within the contract (observed: 0)
        for g, o, lay_g, lay_o in ((sa, os_, M.LAYER_TSDF, oracle_mod.L_TSDF), (da, od, M.LAYER_OCCUPANCY, oracle_mod.L_TSDF)):
            ig = g.block_indices(lay_g); io = o.block_indices(lay_o)
            assert np.array_equal(ig, io) and (len(io) > 0 or g is da)
            bg, _ = g.get_blocks(lay_g, ig)
            field = "log_odds" if g is da else "distance"
            for k, idx in enumerate(io):
                bo = o.get_block(lay_o, idx)
                assert np.abs(bg[k][field] - bo["distance"]).max() <= 1e-4
        assert len(da.block_indices(M.LAYER_OCCUPANCY)) > 0           # the moving box reached the foreground mapper
        # the pair really shared launches: pair kernels in the first mapper's profile, no view-marking launch of its own on most frames
        names = {}
        for k_, v in sa.profile().items():
            names[_short(k_)] = names.get(_short(k_), 0) + v["count"]
        assert names.get("k_mark_view_pair", 0) >= n_frames - 1 and names.get("k_integrate_tsdf_color_pair", 0) >= n_frames - 1, names
        assert names.get("k_mark_view", 0) <= 1, names
        dn = {_short(k_) for k_ in da.profile()}
        assert "k_mark_view" not in dn and "k_integrate_tsdf" not in dn, dn
        for m_ in (sa, sb, da, db):
            m_.close()


def test_depth_pair_falls_back_to_the_two_calls(oracle_mod, hip_lib):
    """Different streams (or the same mapper twice: an argument error) -- the pair is the two calls, nothing else."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M, _lib
    cam = H.SMALL_CAM
    pg = M.default_params()
    a = M.Mapper(pg, block_capacity=1 << 12); b = M.Mapper(pg, block_capacity=1 << 12)          # each on a stream of its own
    ra = M.Mapper(pg, block_capacity=1 << 12); rb = M.Mapper(pg, block_capacity=1 << 12)
    a.set_profiling(True)
    fr = H.frames(3, cam, color=True, stride=9)
    for d, rgb, T in fr:
        d2 = d.copy(); d2[:, : cam[4] // 2] = 0.0
        a.integrate_depth_pair(d, b, d2, T, cam)
        ra.integrate_depth(d, T, cam); rb.integrate_depth(d2, T, cam)
        for m_ in (a, ra):
            m_.integrate_color(rgb, T, cam); m_.update_esdf()
        b.update_esdf(); rb.update_esdf()
    for m_ in (a, b, ra, rb):
        m_.synchronize()
    lay = [("tsdf", M.LAYER_TSDF), ("color", M.LAYER_COLOR), ("esdf", M.LAYER_ESDF)]
    _same(M, a, ra, lay, "fallback a"); _same(M, b, rb, lay, "fallback b")
    assert not ({"k_mark_view_pair", "k_integrate_tsdf_color_pair"} & {_short(k_) for k_ in a.profile()}), list(a.profile())
    d_dev = torch.from_numpy(fr[0][0]).cuda(); Tm = np.ascontiguousarray(np.asarray(fr[0][2], np.float32).reshape(4, 4))
    k = M.Camera(*[float(v) for v in cam[:4]], int(cam[4]), int(cam[5]))
    assert hip_lib.nvbx_integrate_depth_pair(a._h, C.c_void_p(d_dev.data_ptr()), a._h, C.c_void_p(d_dev.data_ptr()), cam[5], cam[4], Tm.ctypes.data_as(C.c_void_p), C.byref(k)) < 0
    for m_ in (a, b, ra, rb):
        m_.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_depth_pair_in_randomised_call_sequences(oracle_mod, hip_lib, seed):
    """Random call patterns around the pair: colour frames or not, ESDF updates on either mapper or not, decay, radius clearing, slice queries, mesh updates,
    a change of the deferral mode, a frame integrated by the two calls in between -- the pair mappers against two mappers driven by the separate calls,
    layers compared at random check points and at the end (bit for bit)."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    rows, cols = cam[5], cam[4]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    rng = np.random.default_rng(100 + seed)
    ps = M.default_params(tsdf_decay_factor=0.8, tsdf_decayed_weight_threshold=0.3)
    pd = M.default_params(projective_layer_type=1, free_region_decay_probability=0.6, occupied_region_decay_probability=0.35)
    lay_s = [("tsdf", M.LAYER_TSDF), ("color", M.LAYER_COLOR), ("esdf", M.LAYER_ESDF)]
    lay_d = [("occupancy", M.LAYER_OCCUPANCY), ("esdf", M.LAYER_ESDF)]
    with torch.cuda.stream(stream):
        sa = M.Mapper(ps, block_capacity=1 << 13, stream=stream.cuda_stream); da = M.Mapper(pd, block_capacity=1 << 12, stream=stream.cuda_stream)
        sb = M.Mapper(ps, block_capacity=1 << 13, stream=stream.cuda_stream); db = M.Mapper(pd, block_capacity=1 << 12, stream=stream.cuda_stream)
        fr = H.frames(10, cam, color=True, stride=7)
        n_pair = 0
        for step in range(36):
            d, rgb, T = fr[int(rng.integers(len(fr)))]
            # the two halves: a vertical split at a random column (foreground = the right part), now and then an empty foreground
            cut = int(rng.integers(cols // 4, cols)) if rng.random() < 0.85 else cols
            bg = d.copy(); bg[:, cut:] = 0.0
            fg = d.copy(); fg[:, :cut] = 0.0
            if rng.random() < 0.85:
                sa.integrate_depth_pair(bg, da, fg, T, cam); n_pair += 1
            else:                                                  # (the pair mappers take a frame by the two calls now and then)
                sa.integrate_depth(bg, T, cam); da.integrate_depth(fg, T, cam)
            sb.integrate_depth(bg, T, cam); db.integrate_depth(fg, T, cam)
            if rng.random() < 0.7:
                sa.integrate_color(rgb, T, cam); sb.integrate_color(rgb, T, cam)
            if rng.random() < 0.7:
                sa.update_esdf(); sb.update_esdf()
            if rng.random() < 0.6:
                da.update_esdf(); db.update_esdf()
            r = rng.random()
            if r < 0.12:
                sa.decay_tsdf(True); sb.decay_tsdf(True)
            elif r < 0.22:
                da.decay_occupancy(); db.decay_occupancy()
            elif r < 0.30:
                c = (float(T[0, 3]), float(T[1, 3]), 1.0)
                sa.clear_outside_radius(c, 2.5); sb.clear_outside_radius(c, 2.5)
            elif r < 0.40:
                ia, _ = sa.esdf_slice_image(); ib, _ = sb.esdf_slice_image()
                assert ia.shape == ib.shape and np.array_equal(ia, ib), (seed, step)
            elif r < 0.48:
                sa.update_color_mesh(); sb.update_color_mesh()
            elif r < 0.54:
                on = bool(rng.integers(2))
                sa.set_color_deferral(on, staged=True); sb.set_color_deferral(on, staged=True)
            if rng.random() < 0.15:
                _same(M, sa, sb, lay_s, ("static", seed, step)); _same(M, da, db, lay_d, ("dynamic", seed, step))
        for m_ in (sa, sb, da, db):
            m_.synchronize()
        _same(M, sa, sb, lay_s, ("static, end", seed)); _same(M, da, db, lay_d, ("dynamic, end", seed))
        assert n_pair > 20 and sa.counters()["capacity_overflow"] == 0
        for m_ in (sa, sb, da, db):
            m_.close()


def test_a_view_of_two_thousand_blocks_takes_the_pipeline_like_the_room(oracle_mod, hip_lib):
    """bench.py --scene hall: the metric's camera and trajectory in a 14 x 12 x 3 m hall -- ~2 400 blocks in view and ~1 200 colour candidates per frame, more
    than the fused launch has workgroups for (its TSDF and colour parts grid-stride, the records taken in runs per XCD with an unpermuted tail), pools that grow twice
    on the way (4 096 -> 16 384 blocks) with work held back.  Eight frames of depth + colour + updateEsdf on a default mapper (two launches per frame) and on one in
    classic order: the same map bit for bit, and the checker's."""
    from isaac_ros_nvblox_amd import mapper as M
    from test_gpu_parity import compare_layer, TOL
    from test_gpu_sequences import check_all
    cam = S.REPLICA_LIKE_CAM
    hall = S.Scene(room_min=(-7.0, -6.0, 0.0), room_max=(7.0, 6.0, 3.0))
    pg = M.default_params(); po = H.copy_params(pg, oracle_mod.OrcParams)
    piped = M.Mapper(pg, block_capacity=1 << 12); classic = M.Mapper(pg, block_capacity=1 << 12); o = oracle_mod.OracleMap(po)
    classic.set_color_deferral(False)
    n_view = []
    for i in range(8):
        T = S.trajectory_pose(i * 9, 200)
        d, rgb = S.render(hall, T, cam)
        for m_ in (classic, piped, o):
            m_.integrate_depth(d, T, cam); m_.integrate_color(rgb, T, cam); m_.update_esdf()
        n_view.append(len(np.asarray(o.last_view()).reshape(-1, 3)))
        assert H.idx_set(piped.last_view()) == H.idx_set(o.last_view())
    assert min(n_view) > 1500, n_view
    for layer, fields in ((M.LAYER_TSDF, ("distance", "weight")), (M.LAYER_COLOR, ("r", "g", "b", "weight"))):
        ia = classic.block_indices(layer); ib = piped.block_indices(layer)
        assert np.array_equal(ia, ib)
        ba, _ = classic.get_blocks(layer, ia); bb, _ = piped.get_blocks(layer, ia)
        for f in fields:
            assert np.array_equal(ba[f], bb[f]), (layer, f)
    piped.update_color_mesh(full=True); o.update_mesh(full=True)
    n_tri = check_all(M, oracle_mod, piped, o)
    assert n_tri > 50000 and piped.counters()["capacity_overflow"] == 0 and piped.capacity >= 1 << 13
