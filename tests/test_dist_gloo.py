"""World-size-2 CPU (gloo) coverage of the N>1 path: the dirty-block all-gather of isaac_ros_nvblox_amd/dist.py and the
camera sharding used by bench.py.  The GPU side of the exchange (nvbx_esdf_dirty_list / nvbx_mark_esdf_dirty) is covered
by tests/test_gpu_multi.py on one GPU with two mappers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaac_ros_nvblox_amd.dist import DirtyBlockExchange, camera_yaw_offset_deg
    ex = DirtyBlockExchange(64, torch.device("cpu"))
    rng = np.random.default_rng(rank)
    n = 10 + 7 * rank
    mine = rng.integers(-20, 20, size=(n, 3)).astype(np.int32)
    ex.idx[:n] = torch.from_numpy(mine); ex.cnt[0] = n
    work = ex.all_gather(async_op=True)      # split-phase, as bench.py uses it around the colour pass
    work.wait()
    got = ex.union_host()
    # expected union computed independently from the known seeds
    want = set()
    for r in range(world):
        rr = np.random.default_rng(r)
        want |= set(map(tuple, rr.integers(-20, 20, size=(10 + 7 * r, 3)).astype(np.int32).tolist()))
    ok = got == want and ex.all_cnt.tolist() == [10 + 7 * r for r in range(world)]
    ok = ok and camera_yaw_offset_deg(rank, world) == 45.0 * rank
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dirty_block_allgather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_exchange_is_identity():
    from isaac_ros_nvblox_amd.dist import DirtyBlockExchange
    ex = DirtyBlockExchange(8, torch.device("cpu"))
    ex.idx[:3] = torch.tensor([[1, 2, 3], [4, 5, 6], [1, 2, 3]], dtype=torch.int32); ex.cnt[0] = 3
    ex.all_gather()
    assert ex.union_host() == {(1, 2, 3), (4, 5, 6)}


class _StubMapper:
    """Stands in for Mapper in the pipelined exchange: exports a per-frame list that encodes (rank, frame), records what is applied.
    late_read: like a mapper with colour deferral on, a deferred union step is only REMEMBERED (the gathered buffer by reference) and read while the
    NEXT frame's depth pass runs -- here: when that frame's message is exported -- so the buffer set must still hold the lists then."""
    def __init__(self, rank, late_read=False):
        self.rank = rank; self.frame = 0; self.applied = []; self.late_read = late_read; self.held = None

    def _read(self, gathered, world, self_rank):
        got = []
        for r in range(world):
            if r == self_rank:
                continue
            c = int(gathered[r, 0, 0]); got += [tuple(v) for v in gathered[r, 1:1 + c].tolist()]
        self.applied.append(got)

    def esdf_dirty_list(self, idx_out, count_out):
        if self.held is not None:            # the held-back union step rides in this frame's depth launches
            self._read(*self.held); self.held = None
        n = 3 + self.rank + (self.frame % 2)
        idx_out[:n] = torch.tensor([[self.rank, self.frame, k] for k in range(n)], dtype=torch.int32)
        count_out[0] = n

    def mark_esdf_dirty_gathered(self, gathered, world, self_rank, max_count, deferred=False):
        if self.late_read and deferred:
            self.held = (gathered, world, self_rank)
        else:
            if self.held is not None:
                self._read(*self.held); self.held = None
            self._read(gathered, world, self_rank)


def _pipelined_worker(rank, world, port, q, late_read=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaac_ros_nvblox_amd.dist import PipelinedDirtyBlockExchange
    ex = PipelinedDirtyBlockExchange(16, torch.device("cpu"))
    m = _StubMapper(rank, late_read)
    n_frames = 5
    for f in range(n_frames):           # bench.py's step: depth, start, finish_previous (deferred), colour, updateEsdf
        m.frame = f
        ex.start(m)
        ex.finish_previous(m, deferred=True)
    ex.drain(m)
    peer = 1 - rank
    want = [[(peer, f, k) for k in range(3 + peer + (f % 2))] for f in range(n_frames)]
    # frame 0's finish_previous had nothing to apply; every frame's peer list is applied exactly once, in order, the last by drain()
    ok = m.applied == want
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("late_read", [False, True], ids=["classic_order", "deferred_mapper_reads_one_depth_pass_later"])
def test_pipelined_exchange_applies_every_list_once_one_frame_late_world2(late_read):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q, late_read)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


# ---------------------------------------------------------------------------------------------- one fused map: measurement exchange
def _fusion_worker(rank, world, port, q, sharded, pipelined=False):
    """Every rank integrates ITS camera through dist.MeasurementFusion (gloo all-gather of the measurement records) into a CPU map (the
    oracle as the rank's mapper); the result must be the map ONE mapper gets from the same cameras in rank order -- bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    from isaac_ros_nvblox_amd import synthetic as S
    from isaac_ros_nvblox_amd.dist import MeasurementFusion
    cam = (80.0, 80.0, 79.5, 59.5, 160, 120)
    p = oracle.default_params(weighting_mode=4, invalid_depth_decay_factor=0.8, max_weight=3.0)
    mine = oracle.OracleMap(p)
    single = oracle.OracleMap(p)                       # the reference: all cameras, in rank order, through one mapper
    from isaac_ros_nvblox_amd.dist import PipelinedMeasurementFusion
    fusion = PipelinedMeasurementFusion(1024, torch.device("cpu"), sharded=sharded) if pipelined else MeasurementFusion(1024, torch.device("cpu"), sharded=sharded)
    core = fusion.f if pipelined else fusion
    sc = S.Scene()
    ok = True
    applied = 0
    for k in range(4 if pipelined else 3):
        frames = []
        for r in range(world):
            T = S.trajectory_pose(k * 11, 200, yaw_offset_deg=45.0 * r)      # overlapping views
            d, _ = S.render(sc, T, cam, color=False)
            d[10:30, 20:60] = 0.0                                             # invalid depth: the decay path is exchanged too
            frames.append((d, T))
        if pipelined:                                                         # bench.py's step order: begin(i), finish_previous() -> frame i-1 applied
            fusion.begin(mine, frames[rank][0], frames[rank][1], cam)
            applied += 1 if fusion.finish_previous(mine) else 0
            ok = ok and applied == k                                          # one frame late
        else:
            fusion.integrate_depth(mine, frames[rank][0], frames[rank][1], cam)
        # only what is used goes to the collective: the payload is max(count) rounded up to 64 records, never the 1024-record buffer
        if not pipelined or k > 0:
            ok = ok and 0 < core.used_records <= core.sent_records <= core.used_records + 63 and core.sent_records % 64 == 0 and core.sent_records < 1024
        for d, T in frames:
            single.integrate_depth(d, T, cam)
    if pipelined:
        ok = ok and fusion.drain(mine) == 1 and fusion.drain(mine) == 0
    ok = ok and core.sent_bytes_total <= 1.25 * core.used_bytes_total + 64 * 4112
    idx_single = single.block_indices(oracle.L_TSDF)
    idx_mine = mine.block_indices(oracle.L_TSDF)
    n_cmp = 0
    if not sharded:
        ok = ok and np.array_equal(idx_single, idx_mine)
    for idx in idx_single:
        owned = (not sharded) or (int(oracle.lib().orc_index_hash(int(idx[0]), int(idx[1]), int(idx[2]))) % world == rank)
        if not owned:
            continue
        a = single.get_block(oracle.L_TSDF, idx); b = mine.get_block(oracle.L_TSDF, idx)
        ok = ok and b is not None and np.array_equal(a["distance"], b["distance"]) and np.array_equal(a["weight"], b["weight"])
        n_cmp += 1
    # ESDF of the fused replica == ESDF of the single map
    if not sharded:
        mine.update_esdf(); single.update_esdf()
        sa, _ = mine.esdf_slice_image(); sb, _ = single.esdf_slice_image()
        ok = ok and sa.shape == sb.shape and np.array_equal(sa, sb)
    q.put((rank, bool(ok), n_cmp, len(idx_single)))
    dist.barrier()
    dist.destroy_process_group()


def _run_fusion(world, sharded, pipelined=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fusion_worker, args=(r, world, port, q, sharded, pipelined)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_measurement_fusion_gives_every_rank_the_single_mapper_map_world2():
    res = _run_fusion(2, sharded=False)
    assert [r[:2] for r in res] == [(0, True), (1, True)] and all(r[2] == r[3] > 300 for r in res)


def test_measurement_fusion_owner_shards_partition_the_single_mapper_map_world3():
    res = _run_fusion(3, sharded=True)
    assert [r[:2] for r in res] == [(0, True), (1, True), (2, True)]
    assert sum(r[2] for r in res) == res[0][3] and min(r[2] for r in res) > 50          # the shards partition the map


def test_pipelined_measurement_fusion_equals_single_mapper_world2():
    """dist.PipelinedMeasurementFusion: payload of frame i-1 in flight while frame i is measured, applied one frame late, the last one
    by drain(); same map as ONE mapper, bit for bit; the bytes handed to the collective follow the used records (<= 1.25 x)."""
    res = _run_fusion(2, sharded=False, pipelined=True)
    assert [r[:2] for r in res] == [(0, True), (1, True)] and all(r[2] == r[3] > 300 for r in res)


def test_pipelined_measurement_fusion_owner_shards_world3():
    res = _run_fusion(3, sharded=True, pipelined=True)
    assert [r[:2] for r in res] == [(0, True), (1, True), (2, True)]
    assert sum(r[2] for r in res) == res[0][3] and min(r[2] for r in res) > 50
