"""Colour deferral / cross-frame pipelining (nvbx_mapper_set_color_deferral): view marking of depth frame i+1 runs in one
launch with the sphere tracing of colour frame i.  Whatever the order of launches, the API must observe call order: a deferred mapper and
a classic one fed the same calls hold bit-identical maps at every point where anything is read."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _equal_maps(M, a, b, tag=""):
    for layer, fields in ((M.LAYER_TSDF, ("distance", "weight")), (M.LAYER_COLOR, ("r", "g", "b", "weight")),
                          (M.LAYER_ESDF, ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site"))):
        ia = a.block_indices(layer); ib = b.block_indices(layer)
        assert np.array_equal(ia, ib), (tag, layer, len(ia), len(ib))
        if len(ia) == 0:
            continue
        ba, _ = a.get_blocks(layer, ia); bb, _ = b.get_blocks(layer, ia)
        for f in fields:
            assert np.array_equal(ba[f], bb[f]), (tag, layer, f)
    sa, aa = a.esdf_slice_image(); sb, ab = b.esdf_slice_image()
    assert sa.shape == sb.shape and np.array_equal(sa, sb) and np.array_equal(aa, ab)


@pytest.mark.parametrize("staged", [True, False], ids=["staged", "zero_copy"])
@pytest.mark.parametrize("cam", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_steady_state_pipeline_equals_classic_and_oracle(oracle_mod, hip_lib, cam, staged):
    """The bench's loop -- depth, colour, updateEsdf per frame -- for 12 frames: every frame but the first takes the pipelined path.  Both forms of
    the deferral: staged (the shipped default: a raw-pointer image is copied first) and zero-copy (opt-in) -- ADVICE r04."""
    from isaac_ros_nvblox_amd import mapper as M
    from test_gpu_parity import compare_layer, TOL
    pg = M.default_params(); po = H.copy_params(pg, oracle_mod.OrcParams)
    classic = M.Mapper(pg, block_capacity=1 << 14); piped = M.Mapper(pg, block_capacity=1 << 14); o = oracle_mod.OracleMap(po)
    classic.set_color_deferral(False)          # (a new mapper defers in the staged form: the reference order is asked for)
    piped.set_color_deferral(True, staged=staged)
    for k, (d, rgb, T) in enumerate(H.frames(12, cam, stride=7)):
        for m_ in (classic, piped, o):
            m_.integrate_depth(d, T, cam); m_.integrate_color(rgb, T, cam); m_.update_esdf()
        assert H.idx_set(piped.last_view()) == H.idx_set(o.last_view())
        if k in (0, 5, 11):                  # (a query flushes the pipeline: the frames in between run pipelined, undisturbed)
            _equal_maps(M, classic, piped)
            assert H.idx_set(piped.last_color_view()) == H.idx_set(o.last_color_view())
    compare_layer(M, piped, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    compare_layer(M, piped, o, M.LAYER_COLOR, oracle_mod.L_COLOR, fields_tol=("weight",), lsb_fields=("r", "g", "b"))
    sg, _ = piped.esdf_slice_image(); so, _ = o.esdf_slice_image()
    assert sg.shape == so.shape and np.abs(sg - so).max() <= TOL
    piped.update_color_mesh(); classic.update_color_mesh()
    ma, mb = piped.mesh(), classic.mesh()
    assert ma.keys() == mb.keys() and all(np.array_equal(ma[k_]["triangles"], mb[k_]["triangles"]) and np.array_equal(ma[k_]["vertices"], mb[k_]["vertices"]) for k_ in ma)
    assert piped.counters()["capacity_overflow"] == 0


def test_deferred_calls_are_replayed_by_every_other_entry_point(oracle_mod, hip_lib):
    """Irregular call patterns: two colour frames in a row, an ESDF update without a colour frame, colour without ESDF, bgra8 colour, a LiDAR
    scan / decay / clearing / mesh / batch right behind a held-back colour frame, switching deferral off with a frame pending, clear()."""
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(tsdf_decay_factor=0.7, tsdf_decayed_weight_threshold=0.2, lidar_max_integration_distance_m=6.0)
    a = M.Mapper(pg, block_capacity=1 << 14); b = M.Mapper(pg, block_capacity=1 << 14)
    a.set_color_deferral(False)
    b.set_color_deferral(True)
    fr = H.frames(16, cam, stride=5)
    lidar = (128, 16, 0.1, -np.deg2rad(20.0), np.deg2rad(20.0))
    Tl = np.eye(4, dtype=np.float32); Tl[:3, 3] = (-1.0, 0.5, 1.0)
    rng_img = S.Scene().raycast(Tl[:3, 3].astype(float), S.lidar_beam_dirs(lidar).reshape(-1, 3)).reshape(16, 128).astype(np.float32)

    def both(fn):
        fn(a); fn(b)
    for k, (d, rgb, T) in enumerate(fr):
        both(lambda m: m.integrate_depth(d, T, cam))
        if k % 4 != 3:
            both(lambda m: m.integrate_color(rgb, T, cam))
        if k % 5 == 1:
            both(lambda m: m.integrate_color(fr[(k + 3) % 16][1], fr[(k + 3) % 16][2], cam))          # a second camera's colour frame
        if k % 3 != 2:
            both(lambda m: m.update_esdf())
        if k == 2:
            both(lambda m: m.integrate_lidar_depth(rng_img, Tl, lidar))
        if k == 4:
            both(lambda m: m.decay_tsdf(True))
        if k == 6:
            both(lambda m: m.clear_outside_radius((float(T[0, 3]), float(T[1, 3]), 1.0), 3.0))
        if k == 7:
            both(lambda m: m.update_color_mesh())
        if k == 8:
            both(lambda m: m.integrate_depth_batch([fr[1][0], fr[9][0]], [fr[1][2], fr[9][2]], cam))
            both(lambda m: m.integrate_color_batch([fr[1][1], fr[9][1]], [fr[1][2], fr[9][2]], cam))
        if k == 10:
            b.set_color_deferral(False)
        if k == 12:
            b.set_color_deferral(True)
        if k % 4 == 1 or k == 15:
            _equal_maps(M, a, b, "frame %d" % k)
    both(lambda m: m.update_esdf())
    _equal_maps(M, a, b)
    assert len(a.block_indices(M.LAYER_COLOR)) > 50
    # clear() with a frame held back: the frame is dropped with the map
    d, rgb, T = fr[0]
    both(lambda m: m.integrate_depth(d, T, cam)); both(lambda m: m.integrate_color(rgb, T, cam))
    both(lambda m: m.clear())
    assert b.num_blocks(M.LAYER_TSDF) == 0
    both(lambda m: m.integrate_depth(d, T, cam)); both(lambda m: m.integrate_color(rgb, T, cam)); both(lambda m: m.update_esdf())
    _equal_maps(M, a, b)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fused_colour_tsdf_launch_under_irregular_calls(oracle_mod, hip_lib, seed):
    """Two launches per frame: view marking (i+1) || sphere tracing (i) || colour candidates (i) || ESDF site marking (i), then
    TSDF update (i+1) || colour integration (i) || distance transform (i).  Random call patterns on a plain camera mapper (no LiDAR: the fused
    launch stays eligible) -- colour frames or ESDF updates left out, two updates back to back, bgra8 frames, decay / clearing / meshing /
    queries at random points, a pool that grows in mid-pipeline; a classic mapper fed the same calls must hold the same map bit for bit
    wherever anything is read."""
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    rng = np.random.default_rng(100 + seed)
    pg = M.default_params(tsdf_decay_factor=0.8, tsdf_decayed_weight_threshold=0.05)
    a = M.Mapper(pg, block_capacity=1024); b = M.Mapper(pg, block_capacity=1024)      # (the pool grows in mid-sequence)
    a.set_color_deferral(False)
    b.set_color_deferral(True); b.set_profiling(True)
    fr = H.frames(40, cam, stride=5)

    def both(fn):
        fn(a); fn(b)
    for k, (d, rgb, T) in enumerate(fr):
        dd = d if k % 5 else np.round(d * 1000.0).astype(np.uint16)       # (every 5th depth frame: 16-bit millimetres, converted in the kernels' fetch)
        both(lambda m: m.integrate_depth(dd, T, cam))
        if rng.random() < 0.85:
            img = rgb if k % 7 else np.ascontiguousarray(np.concatenate([rgb[..., ::-1], np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2))     # (every 7th frame: bgra8)
            both(lambda m: m.integrate_color(img, T, cam))
        if rng.random() < 0.8:
            both(lambda m: m.update_esdf())
        r = rng.random()
        if r < 0.06:
            both(lambda m: m.decay_tsdf(True))
        elif r < 0.12:
            both(lambda m: m.clear_outside_radius((float(T[0, 3]), float(T[1, 3]), 1.0), 3.5))
        elif r < 0.18:
            both(lambda m: m.update_color_mesh())
        elif r < 0.24:
            both(lambda m: m.update_esdf())                  # two updates back to back
        elif r < 0.30:
            sa, _ = a.esdf_slice_image(); sb, _ = b.esdf_slice_image()      # a query: flushes the pipeline
            assert np.array_equal(sa, sb), k
        if k % 9 == 8:
            _equal_maps(M, a, b, "seed %d frame %d" % (seed, k))
    both(lambda m: m.update_esdf())
    _equal_maps(M, a, b, "seed %d end" % seed)
    prof = b.profile()
    names = " ".join(prof.keys())
    assert "k_integrate_tsdf_color<Img, PixRgb8" in names and "k_integrate_tsdf_color<Img, PixBgra8" in names, names[:600]      # the fused launch has run, both encodings
    assert "k_grow_commit" in names, names[:600]                                                                                  # ... and the pool has grown under it
    assert b.counters()["capacity_overflow"] == 0 and len(a.block_indices(M.LAYER_COLOR)) > 50


@pytest.mark.parametrize("ncam", [2, 4, 8])
def test_camera_batches_take_the_pipeline_too(oracle_mod, hip_lib, ncam):
    """nvbx_integrate_depth_batch / _color_batch with deferral on: the colour BATCH is held back and the next depth batch carries it out in two
    launches (k_mark_view<.., 8> with the batch's sphere tracing / candidates / ESDF marking riding, then k_integrate_tsdf_color<.., 8>).
    Same calls on a classic mapper: the same map, views and mesh, bit for bit; irregular steps (a batch without colour, an ESDF update left
    out, a single frame between batches, decay) are replayed in call order."""
    from isaac_ros_nvblox_amd import mapper as M
    cam = H.SMALL_CAM
    pg = M.default_params(tsdf_decay_factor=0.8, tsdf_decayed_weight_threshold=0.05)
    a = M.Mapper(pg, block_capacity=1 << 14); b = M.Mapper(pg, block_capacity=1 << 14)
    a.set_color_deferral(False)
    b.set_color_deferral(True); b.set_profiling(True)
    fr = H.frames(14 * ncam, cam, stride=2)

    def both(fn):
        fn(a); fn(b)
    for k in range(14):
        grp = fr[k * ncam:(k + 1) * ncam]
        ds, cs, Ts = [g[0] for g in grp], [g[1] for g in grp], [g[2] for g in grp]
        both(lambda m: m.integrate_depth_batch(ds, Ts, cam))
        if k != 5:
            both(lambda m: m.integrate_color_batch(cs, Ts, cam))
        if k not in (3, 9):
            both(lambda m: m.update_esdf())
        if k == 6:
            both(lambda m: m.integrate_depth(ds[0], Ts[0], cam)); both(lambda m: m.integrate_color(cs[0], Ts[0], cam))      # a single frame between batches
        if k == 10:
            both(lambda m: m.decay_tsdf(True))
        if k in (2, 7, 13):
            _equal_maps(M, a, b, "%d cameras step %d" % (ncam, k))
            assert H.idx_set(a.last_view()) == H.idx_set(b.last_view()) and H.idx_set(a.last_color_view()) == H.idx_set(b.last_color_view())
    both(lambda m: m.update_esdf())
    _equal_maps(M, a, b, "%d cameras end" % ncam)
    both(lambda m: m.update_color_mesh())
    ma, mb = a.mesh(), b.mesh()
    assert ma.keys() == mb.keys() and all(np.array_equal(ma[k_]["triangles"], mb[k_]["triangles"]) and np.array_equal(ma[k_]["vertices"], mb[k_]["vertices"]) for k_ in ma)
    prof = b.profile()
    fused = sum(v["count"] for k_, v in prof.items() if "k_integrate_tsdf_color" in k_)
    assert fused >= 7, {k_: v["count"] for k_, v in prof.items()}          # (of 15 depth launches: the ones that followed a colour batch directly)


@pytest.mark.parametrize("cam", [H.SMALL_CAM, S.REPLICA_LIKE_CAM], ids=["160x120", "640x480"])
def test_dynamic_mapping_frame_keeps_the_pipeline(oracle_mod, hip_lib, cam):
    """The dynamic-mapping frame (MappingType::kDynamic: detect dynamics -> clean the mask -> split the depth image -> static mapper with a
    freespace layer + occupancy mapper -> colour -> two ESDF updates, decay every 6th frame) starts with nvbx_detect_dynamics on the static
    mapper.  That call reads TSDF voxels and the freespace layer only, so it leaves the held-back colour frame / ESDF update / distance
    transform held back and the next integrateDepth carries them out in pipelined order.  Same calls on classic mappers: identical masks,
    split depth images, maps (TSDF, colour, ESDF, freespace) and occupancy, bit for bit.  (Stream-ordered host, like bench.py: mappers and
    torch share one stream, nothing in the loop waits for the GPU.)"""
    from isaac_ros_nvblox_amd import mapper as M
    import torch
    rows, cols = cam[5], cam[4]          # (640x480: the size profiles/*_bench_decay.json times this frame at)
    min_component = 40 if cols == 160 else 640
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    fs = dict(projective_layer_type=2, max_integration_distance_m=5.0, invalid_depth_decay_factor=0.8, tsdf_decay_factor=0.95,
              min_duration_since_occupied_for_freespace_ms=250)
    occ = dict(projective_layer_type=1, max_integration_distance_m=5.0)
    with torch.cuda.stream(stream):
        mk = lambda prm, cap: M.Mapper(M.default_params(**prm), block_capacity=cap, stream=stream.cuda_stream)
        gs_a, gd_a, gs_b, gd_b = mk(fs, 1 << 14), mk(occ, 1 << 13), mk(fs, 1 << 14), mk(occ, 1 << 13)
        gs_a.set_color_deferral(False); gd_a.set_color_deferral(False)
        gs_b.set_color_deferral(True); gs_b.set_profiling(True)
        gd_b.set_color_deferral(True); gd_b.set_profiling(True)       # (no colour on the occupancy mapper: its updateEsdf alone is held back and carried)
        eye = np.eye(4, dtype=np.float32)
        kept = []
        t = 0
        for i in range(20):
            sc = S.redwood_like_scene(i * 6)                       # a box translating through the room
            T = S.trajectory_pose(i * 3, 200, radius=1.2, height=1.4)
            d, rgb = S.render(sc, T, cam, max_range=5.0)
            d_dev = torch.from_numpy(d).to(dev); rgb_dev = torch.from_numpy(rgb).to(dev)
            outs = []
            for gs, gd in ((gs_a, gd_a), (gs_b, gd_b)):
                mask = torch.empty((rows, cols), dtype=torch.uint8, device=dev); un = torch.empty((rows, cols), dtype=torch.float32, device=dev); ma = torch.empty_like(un)
                gs.detect_dynamics_into(d_dev, T, cam, 5.0, mask)
                raw = mask.clone()
                gs.remove_small_components_inplace(mask, min_component)
                gs.split_depth_by_mask_into(d_dev, mask, eye, cam, cam, 0.25, un, ma)
                gs.set_time_ms(t)
                gs.integrate_depth(un, T, cam); gd.integrate_depth(ma, T, cam)
                gs.integrate_color(rgb_dev, T, cam)
                gs.update_esdf(); gd.update_esdf()
                if i % 6 == 5:
                    gs.decay_tsdf(True); gd.decay_occupancy()
                outs.append((raw, mask, un, ma))
            kept.append((outs, d_dev, rgb_dev))       # (device images stay alive: the colour frame is held back until the next integrateDepth)
            t += 100
            if i % 5 == 4 or i == 19:
                for outs_, _, _ in kept:
                    for x, y in zip(outs_[0], outs_[1]):
                        assert torch.equal(x, y), i
                kept = kept[-1:]
        n_dyn = int(kept[-1][0][0][1].sum().item())
        _equal_maps(M, gs_a, gs_b, "static mapper")
        ia, ib = gs_a.block_indices(M.LAYER_FREESPACE), gs_b.block_indices(M.LAYER_FREESPACE)
        assert np.array_equal(ia, ib) and len(ia) > 50
        fa, _ = gs_a.get_blocks(M.LAYER_FREESPACE, ia); fb, _ = gs_b.get_blocks(M.LAYER_FREESPACE, ia)
        for f in ("last_occupied_timestamp_ms", "consecutive_occupancy_duration_ms", "is_high_confidence_freespace", "initialized"):
            assert np.array_equal(fa[f], fb[f]), f
        oa, ob = gd_a.block_indices(M.LAYER_OCCUPANCY), gd_b.block_indices(M.LAYER_OCCUPANCY)
        assert np.array_equal(oa, ob)
        if len(oa):
            ba, _ = gd_a.get_blocks(M.LAYER_OCCUPANCY, oa); bb, _ = gd_b.get_blocks(M.LAYER_OCCUPANCY, oa)
            assert np.array_equal(ba["log_odds"], bb["log_odds"])
        prof = gs_b.profile()
        n_trace = sum(v["count"] for k_, v in prof.items() if "k_sphere_trace" in k_)
        n_mark = sum(v["count"] for k_, v in prof.items() if "k_mark_view" in k_)
        pd = gd_b.profile()
        assert sum(v["count"] for k_, v in pd.items() if "k_esdf_mark" in k_) <= 8 and sum(v["count"] for k_, v in pd.items() if "k_integrate_tsdf_color" in k_) >= 10, {k_: v["count"] for k_, v in pd.items()}
        assert n_mark >= 20 and n_trace <= 8, ({k_: v["count"] for k_, v in prof.items()}, n_dyn)      # most colour frames' sphere tracing rode in a view-marking launch
