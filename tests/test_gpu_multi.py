"""One-camera-per-GPU logic exercised on ONE GPU with two mappers: export of the dirty block list, the union step and
the ESDF sweep over the union -- compared with two oracle maps doing the same exchange on the host."""
import numpy as np
import pytest

import helpers as H
from isaac_ros_nvblox_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_dirty_list_exchange_matches_oracle(oracle_mod, hip_lib):
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    from isaac_ros_nvblox_amd.dist import DirtyBlockExchange
    pg = M.default_params(); po = H.copy_params(pg, oracle_mod.OrcParams)
    gs = [M.Mapper(pg, block_capacity=1 << 13) for _ in range(2)]
    os_ = [oracle_mod.OracleMap(po) for _ in range(2)]
    dev = torch.device("cuda", 0)
    exs = [DirtyBlockExchange(4096, dev) for _ in range(2)]
    colour = {}
    view_bufs = [torch.zeros((4097, 3), dtype=torch.int32, device=dev) for _ in range(2)]
    for step in range(4):
        for r in range(2):
            d, rgb, T = H.frames(1, H.SMALL_CAM, start=step * 12, color=True, yaw_offset_deg=45.0 * r)[0]
            if step == 3:
                gs[r].set_view_export(view_bufs[r])          # the depth pass writes its block list itself (no export launch)
            gs[r].integrate_depth(d, T, H.SMALL_CAM); os_[r].integrate_depth(d, T, H.SMALL_CAM)
            colour[r] = (rgb, T)
            if step == 3:
                gs[r].synchronize()
                nvw = int(view_bufs[r][0, 0].item())
                assert H.idx_set(view_bufs[r][1:1 + nvw].cpu().numpy()) == H.idx_set(os_[r].last_view()) and nvw == len(os_[r].last_view())
                gs[r].set_view_export(None)
        # export
        lists = []
        for r in range(2):
            gs[r].esdf_dirty_list(exs[r].idx, exs[r].cnt); gs[r].synchronize()
            n = int(exs[r].cnt.item())
            got = H.idx_set(exs[r].idx[:n].cpu().numpy())
            want = H.idx_set(os_[r].esdf_dirty_list())
            assert got == want and n == len(want)
            lists.append(os_[r].esdf_dirty_list())
        # union: each mapper marks the peer's list
        for r in range(2):
            p = 1 - r
            if step == 0:
                gs[r].mark_esdf_dirty(exs[p].idx, exs[p].cnt, 4096)
            elif step == 1:      # one launch for all peers: a 2-rank gathered buffer, own rank skipped; lookup + marking fused
                gathered = torch.stack([exs[0].buf, exs[1].buf])
                gs[r].mark_esdf_dirty_gathered(gathered, 2, r, 4096)
            else:                # the form bench.py uses: held back, carried by the next integrateColor launch (step 2) or
                gathered = torch.stack([exs[0].buf, exs[1].buf])     # launched first thing by updateEsdf (step 3: no colour frame)
                gs[r].mark_esdf_dirty_gathered(gathered, 2, r, 4096, deferred=True)
                if step == 2:
                    gs[r].integrate_color(colour[r][0], colour[r][1], H.SMALL_CAM); os_[r].integrate_color(colour[r][0], colour[r][1], H.SMALL_CAM)
            os_[r].mark_esdf_dirty(lists[p])
            gs[r].update_esdf(); os_[r].update_esdf()
            ig, ag = gs[r].esdf_slice_image(1000.0); io, ao = os_[r].esdf_slice_image(1000.0)
            assert ig.shape == io.shape and np.array_equal(ag, ao) and np.abs(ig - io).max() <= 1e-4
            assert gs[r].counters()["esdf_columns_marked"] > 0


def test_measurement_exchange_equals_one_mapper_batch(oracle_mod, hip_lib):
    """nvbx_measure_depth / nvbx_apply_measurements: three 'ranks' on one GPU (three mappers, buffers concatenated by hand instead of an
    all-gather): every rank's map == ONE mapper integrating the three cameras as a batch (bit for bit) == the oracle fed sequentially."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    from test_gpu_parity import compare_layer
    world, stride = 3, 1024
    cam = H.SMALL_CAM
    pg = M.default_params(weighting_mode=4, invalid_depth_decay_factor=0.8)
    ranks = [M.Mapper(pg, block_capacity=1 << 13) for _ in range(world)]
    single = M.Mapper(pg, block_capacity=1 << 13)
    o = oracle_mod.OracleMap(H.copy_params(pg, oracle_mod.OrcParams))
    all_buf = torch.zeros((world, stride, M.Mapper.MEAS_BLOCK_BYTES), dtype=torch.uint8, device="cuda:0")
    all_cnt = torch.zeros((world,), dtype=torch.int32, device="cuda:0")
    sc = S.Scene()
    for k in range(3):
        fr = []
        for r in range(world):
            T = S.trajectory_pose(k * 11, 200, yaw_offset_deg=45.0 * r)
            d, _ = S.render(sc, T, cam, color=False)
            d[10:30, 20:60] = 0.0
            fr.append((d, T))
        for r in range(world):
            ranks[r].measure_depth(fr[r][0], fr[r][1], cam, all_buf[r], all_cnt[r:r + 1])
            ranks[r].synchronize()                      # (the three mappers have their own streams here; dist.MeasurementFusion orders them with events)
        for r in range(world):
            ranks[r].apply_measurements(all_buf, all_cnt)
            ranks[r].synchronize()
        single.integrate_depth_batch([d for d, _ in fr], [T for _, T in fr], cam)
        for d, T in fr:
            o.integrate_depth(d, T, cam)
    idx = single.block_indices(M.LAYER_TSDF)
    bs, _ = single.get_blocks(M.LAYER_TSDF, idx)
    for r in range(world):
        assert np.array_equal(ranks[r].block_indices(M.LAYER_TSDF), idx)
        br, _ = ranks[r].get_blocks(M.LAYER_TSDF, idx)
        assert np.array_equal(br["distance"], bs["distance"]) and np.array_equal(br["weight"], bs["weight"])
    n, _ = compare_layer(M, ranks[1], o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 300
    # the fused replicas run the rest of the path like any map: ESDF slices agree
    for m_ in ranks + [single]:
        m_.update_esdf()
    s0, _ = single.esdf_slice_image()
    for r in range(world):
        sr, _ = ranks[r].esdf_slice_image()
        assert sr.shape == s0.shape and np.array_equal(sr, s0)
    # owner filter: rank 2's shard of the same map
    shard = M.Mapper(pg, block_capacity=1 << 13)
    shard.apply_measurements(all_buf, all_cnt, owner_mod=world, owner_rank=2)      # (the last frame's buffers)
    bi = shard.block_indices(M.LAYER_TSDF)
    b, _ = shard.get_blocks(M.LAYER_TSDF, bi)
    assert len(bi) > 20 and all(int(oracle_mod.lib().orc_index_hash(int(x), int(y), int(z))) % world == 2 for x, y, z in bi.tolist())


def test_pipelined_measurement_fusion_on_the_hip_mapper(oracle_mod, hip_lib):
    """dist.PipelinedMeasurementFusion driving the HIP mapper (world 1: the collectives are local copies, everything else -- the sized
    payload view with stride = max(count) rounded to 64 records instead of the 1024-record buffer, two alternating measurement buffers,
    the count's asynchronous D2H + event, apply one frame late, colour + ESDF of the applied frame, drain) is the multi-GPU code path.
    The fused mapper must equal a plain mapper fed integrateDepth / integrateColor / updateEsdf frame by frame, bit for bit."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M
    from isaac_ros_nvblox_amd.dist import PipelinedMeasurementFusion
    from test_gpu_parity import compare_layer
    cam = H.SMALL_CAM
    pg = M.default_params(invalid_depth_decay_factor=0.8)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        fused = M.Mapper(pg, block_capacity=1 << 13, stream=stream.cuda_stream)
        plain = M.Mapper(pg, block_capacity=1 << 13)
        o = oracle_mod.OracleMap(H.copy_params(pg, oracle_mod.OrcParams))
        mf = PipelinedMeasurementFusion(1024, torch.device("cuda", 0))
        fr = H.frames(6, cam, stride=9)
        prev = None
        for k, (d, rgb, T) in enumerate(fr):
            mf.begin(fused, d, T, cam)
            if mf.finish_previous(fused):
                fused.integrate_color(prev[1], prev[2], cam); fused.update_esdf()
                assert 0 < mf.f.used_records <= mf.f.sent_records <= mf.f.used_records + 63 and mf.f.sent_records < 1024
            prev = (d, rgb, T)
            plain.integrate_depth(d, T, cam); plain.integrate_color(rgb, T, cam); plain.update_esdf()
            o.integrate_depth(d, T, cam); o.integrate_color(rgb, T, cam); o.update_esdf()
        assert mf.drain(fused) == 1
        fused.integrate_color(prev[1], prev[2], cam); fused.update_esdf()
        assert mf.f.sent_bytes_total <= 1.25 * mf.f.used_bytes_total + 64 * 4112
    idx = plain.block_indices(M.LAYER_TSDF)
    assert np.array_equal(fused.block_indices(M.LAYER_TSDF), idx) and len(idx) > 300
    for layer, fields in ((M.LAYER_TSDF, ("distance", "weight")), (M.LAYER_COLOR, ("r", "g", "b", "weight"))):
        ia = plain.block_indices(layer)
        assert np.array_equal(fused.block_indices(layer), ia)
        a, _ = plain.get_blocks(layer, ia); b, _ = fused.get_blocks(layer, ia)
        for f in fields:
            assert np.array_equal(a[f], b[f]), (layer, f)
    sa, aa = plain.esdf_slice_image(); sb, ab = fused.esdf_slice_image()
    assert sa.shape == sb.shape and np.array_equal(sa, sb) and np.array_equal(aa, ab)
    n, _ = compare_layer(M, fused, o, M.LAYER_TSDF, oracle_mod.L_TSDF, fields_tol=("distance", "weight"))
    assert n > 300
