"""One-camera-per-GPU sharding: all-gather of updated block indices before the ESDF sweep.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU in the tests).
The exchange is ONE fixed-size all-gather per frame: each rank contributes a packed int32 buffer [1 + max_blocks, 3]
whose row 0 holds the count and rows 1.. the block indices (<= 48 KiB at max_blocks = 4096, 8 ranks: 384 KiB gathered).
The message is latency-bound on the point-to-point xGMI links (one RCCL launch, no host copy, no sync, the count never
leaves the device), so it is started asynchronously right after the depth pass and joined before the ESDF sweep: the
colour integration of the same frame runs while the collective is in flight.
The reference has no multi-GPU path at all (nvblox_ros/include/nvblox_ros/nvblox_node.hpp:298-332: <=4 cameras share
one queue on one GPU), so this is new design.
"""
import torch
import torch.distributed as dist


class DirtyBlockExchange:
    """Pre-allocated buffers + the per-frame exchange. `device` may be a CPU device (gloo tests)."""

    def __init__(self, max_blocks, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.max_blocks = int(max_blocks)
        self.buf = torch.zeros((self.max_blocks + 1, 3), dtype=torch.int32, device=device)        # row 0 = (count, 0, 0)
        self.all_buf = torch.zeros((self.world, self.max_blocks + 1, 3), dtype=torch.int32, device=device)
        # views (no copies): what the C-ABI reads / writes
        self.cnt = self.buf[0, 0:1]
        self.idx = self.buf[1:]
        self.all_cnt = self.all_buf[:, 0, 0]
        self.all_idx = self.all_buf[:, 1:]

    def all_gather(self, async_op=False):
        """all-gather self.buf into self.all_buf (local copy when world == 1).  Returns the Work handle if async_op."""
        if self.world == 1:
            self.all_buf[0].copy_(self.buf)
            return None
        return dist.all_gather_into_tensor(self.all_buf.view(-1, 3), self.buf, group=self.group, async_op=async_op)

    # -- stream ordering.  The mapper writes self.buf / reads self.all_buf on ITS stream (Mapper(stream=None) owns a non-blocking
    #    stream); the collective runs on torch's current stream (RCCL's own stream is ordered behind it by torch).  When the two
    #    differ, events order them; when the mapper was created on torch's current stream there is nothing to do.
    def _streams(self, mapper):
        if not self.buf.is_cuda or not hasattr(mapper, "torch_stream"):
            return None, None
        cur = torch.cuda.current_stream(self.buf.device)
        ms = mapper.torch_stream()
        if ms.cuda_stream == cur.cuda_stream:
            return None, None
        return ms, cur

    # -- split-phase exchange (bench.py): start after integrateDepth, finish before updateEsdf
    def start(self, mapper, export=True):
        """Export this GPU's dirty TSDF block indices (device kernel, count stays on the device) and launch the all-gather.
        export=False: the mapper has already written the message (Mapper.set_view_export(self.buf) before integrate_depth)."""
        if export:
            mapper.esdf_dirty_list(self.idx, self.cnt)
        ms, cur = self._streams(mapper)
        if ms is not None:
            cur.wait_stream(ms)              # the collective reads self.buf only after the mapper's stream has written it
        return self.all_gather(async_op=True)

    def finish(self, mapper, work, deferred=False):
        """Join the all-gather, then mark every peer's blocks ESDF-dirty locally (count read on the device).  deferred: the
        marking rides in the mapper's next integrate_color launch instead of a launch of its own."""
        if work is not None:
            work.wait()                      # orders torch's current stream behind the collective
        ms, cur = self._streams(mapper)
        if ms is not None:
            ms.wait_stream(cur)              # ... and the mapper's stream behind torch's: the union step reads all_buf after it landed
        if self.world > 1:      # one launch for all peers' lists (not one per peer)
            if deferred:
                mapper.mark_esdf_dirty_gathered(self.all_buf, self.world, self.rank, self.max_blocks, deferred=True)
            else:
                mapper.mark_esdf_dirty_gathered(self.all_buf, self.world, self.rank, self.max_blocks)

    def exchange(self, mapper):
        self.finish(mapper, self.start(mapper))

    def union_host(self):
        """Host-side union of the gathered lists (tests / diagnostics only; synchronises)."""
        cnt = self.all_cnt.cpu().tolist(); idx = self.all_idx.cpu()
        out = set()
        for r in range(self.world):
            out |= set(map(tuple, idx[r, :min(cnt[r], self.max_blocks)].tolist()))
        return out


class PipelinedDirtyBlockExchange:
    """The same exchange, software-pipelined by one frame: frame i's all-gather is started after its depth pass and joined
    just before frame i+1's ESDF update, so the collective has a whole frame of GPU work (colour(i), ESDF(i), depth(i+1),
    colour(i+1)) to hide its latency behind instead of one colour pass.  Every list is still applied exactly once, one ESDF
    update later than in the unpipelined form -- for the peers' lists that is immaterial (each GPU sweeps its OWN TSDF over
    the union; a peer's update of a block changes nothing in the local layer).  THREE buffer sets rotate: the set whose
    collective is in flight is never written, and neither is the set a mapper with colour deferral still reads -- there the
    union step of frame i's lists rides in a launch of integrateDepth(i + 2) (the fused TSDF-update launch),
    i.e. while frame i + 1's collective is in flight and frame i + 2's message is being written.  drain() joins the last one
    (call it before reading results / timing)."""

    N_SLOTS = 3

    def __init__(self, max_blocks, device, group=None):
        self.slots = [DirtyBlockExchange(max_blocks, device, group) for _ in range(self.N_SLOTS)]
        self.world = self.slots[0].world
        self.frame = 0
        self.pending = None          # (slot, work) of the frame whose lists have not been applied yet
        self.started = None          # (slot, work) of the current frame

    def before_depth(self, mapper):
        """Optional, before integrateDepth of the current frame: the depth pass itself writes the message (no export launch)."""
        mapper.set_view_export(self.slots[self.frame % self.N_SLOTS].buf)
        self.registered = True

    def start(self, mapper):
        """After integrateDepth of the current frame."""
        slot = self.slots[self.frame % self.N_SLOTS]
        self.started = (slot, slot.start(mapper, export=not getattr(self, "registered", False)))
        self.registered = False
        self.frame += 1

    def finish_previous(self, mapper, deferred=False):
        """Between integrateDepth and updateEsdf of the current frame: apply the PREVIOUS frame's gathered lists; the current
        frame's stay in flight.  deferred=True (call it BEFORE integrateColor): the marking rides in that colour launch."""
        if self.pending is not None:
            slot, work = self.pending
            slot.finish(mapper, work, deferred=deferred)
        self.pending, self.started = self.started, None

    def drain(self, mapper):
        if self.pending is not None:
            slot, work = self.pending
            slot.finish(mapper, work)
            self.pending = None


class MeasurementFusion:
    """One fused map from one camera per GPU (SURVEY.md 8e option B, made exact -- include/nvblox_hip.h "measurement exchange").

    Per frame: mapper.measure_depth (this rank's view calculation + projection, the sharded part) -> all-gather of the record COUNTS
    (4 B per rank) -> ONE all_gather_into_tensor of the first n records of every rank's buffer, n = max(count) rounded up to 64 records
    (only what is used goes over the links: ~300 blocks in view = 320 records = 1.3 MB per rank, not the 4.2 MB the buffer is sized
    for) -> mapper.apply_measurements on every rank, in rank order, with stride n.  Every rank then holds the SAME map, bit-identical
    to a single mapper integrating the cameras in rank order; `sharded=True` keeps only the blocks this rank owns (Index3DHash mod
    world).  Sizing the payload needs the counts on the HOST: this unpipelined form waits for them (one small D2H + sync per frame);
    PipelinedMeasurementFusion below hides that wait and the payload's wire time behind a frame of GPU work.
    The mapper may be the HIP Mapper (device tensors, RCCL) or any object with the same two methods (CPU tensors, gloo: tests).
    `sent_records` / `used_records`: records per rank handed to the last payload collective / the largest count among the ranks."""

    BLOCK_BYTES = 4112
    ROUND = 64

    def __init__(self, stride_blocks, device, group=None, sharded=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stride = int(stride_blocks)
        self.sharded = bool(sharded)
        self.device = device
        self.buf = torch.zeros((self.stride, self.BLOCK_BYTES), dtype=torch.uint8, device=device)
        self.cnt = torch.zeros((1,), dtype=torch.int32, device=device)
        self.all_flat = torch.zeros((self.world * self.stride * self.BLOCK_BYTES,), dtype=torch.uint8, device=device)
        self.all_cnt = torch.zeros((self.world,), dtype=torch.int32, device=device)
        self.sent_records = 0; self.used_records = 0; self.sent_bytes_total = 0; self.used_bytes_total = 0

    def _order(self, mapper, first, second):
        """event ordering between the mapper's stream and torch's current stream when they differ (see DirtyBlockExchange._streams)"""
        if not self.buf.is_cuda or not hasattr(mapper, "torch_stream"):
            return
        cur = torch.cuda.current_stream(self.buf.device); ms = mapper.torch_stream()
        if ms.cuda_stream == cur.cuda_stream:
            return
        (cur if first == "mapper" else ms).wait_stream(ms if first == "mapper" else cur)

    # -- the three pieces of the exchange (PipelinedMeasurementFusion re-orders them across frames)
    def gather_counts(self, cnt, all_cnt):
        if self.world == 1:
            all_cnt.copy_(cnt)
        else:
            dist.all_gather_into_tensor(all_cnt, cnt, group=self.group)

    def sized(self, counts_host):
        """records per rank the payload collective carries: max(count) rounded up to ROUND, at least ROUND, at most the buffer"""
        used = max(0, min(self.stride, max(int(c) for c in counts_host)))
        n = min(self.stride, max(self.ROUND, -(-used // self.ROUND) * self.ROUND))
        return n, used

    def gather_payload(self, buf, n, async_op=False):
        """all-gather the first n records of every rank's buffer into all_flat viewed as [world, n, 4112]"""
        out = self.all_flat[: self.world * n * self.BLOCK_BYTES].view(self.world, n, self.BLOCK_BYTES)
        self.sent_records = n; self.sent_bytes_total += n * self.BLOCK_BYTES
        if self.world == 1:
            out[0].copy_(buf[:n]); return out, None
        work = dist.all_gather_into_tensor(out.view(-1, self.BLOCK_BYTES), buf[:n], group=self.group, async_op=async_op)
        return out, (work if async_op else None)

    def apply(self, mapper, gathered, all_cnt):
        if self.sharded:
            mapper.apply_measurements(gathered, all_cnt, self.world, self.rank)
        else:
            mapper.apply_measurements(gathered, all_cnt)

    def integrate_depth(self, mapper, depth, T_L_C, cam):
        """The multi-GPU form of MultiMapper::integrateDepth: collective, every rank calls it with its own camera frame."""
        mapper.measure_depth(depth, T_L_C, cam, self.buf, self.cnt)
        self._order(mapper, "mapper", "torch")             # the collectives read the buffers only after the mapper's stream has written them
        self.gather_counts(self.cnt, self.all_cnt)
        n, used = self.sized(self.all_cnt.cpu().tolist())  # (synchronises: the payload is sized on the host)
        self.used_records = used; self.used_bytes_total += used * self.BLOCK_BYTES
        gathered, _ = self.gather_payload(self.buf, n)
        self._order(mapper, "torch", "mapper")
        self.apply(mapper, gathered, self.all_cnt)


class PipelinedMeasurementFusion:
    """MeasurementFusion software-pipelined by one frame, like PipelinedDirtyBlockExchange: nothing waits for a collective it has just
    started.  Per step i the caller runs

        begin(mapper, depth_i, T_i, cam)     1. payload all-gather of frame i-1 STARTS (sized from its counts, which reached the host a
                                                frame ago) -- 2. measure_depth of frame i runs while it is in flight -- 3. the count
                                                all-gather of frame i + an asynchronous D2H of the counts are enqueued
        if finish_previous(mapper): ...      4. the payload of frame i-1 is joined and applied (rank order); returns True if a frame was
                                                applied -- the caller then runs integrateColor / updateEsdf OF FRAME i-1
        drain(mapper)                        after the last frame: its payload is gathered and applied

    measure_depth does not read voxel values (it allocates the blocks in view and samples the depth image), so measuring frame i
    before frame i-1 is applied changes nothing: the map after every apply is bit-identical to the unpipelined form and to ONE mapper
    integrating the cameras in rank order.  Two measurement buffers alternate (the collective in flight reads one while measure_depth
    writes the other)."""

    def __init__(self, stride_blocks, device, group=None, sharded=False):
        self.f = MeasurementFusion(stride_blocks, device, group, sharded)
        self.world = self.f.world
        cuda = self.f.buf.is_cuda
        self.bufs = [self.f.buf, torch.zeros_like(self.f.buf)]
        self.cnts = [self.f.cnt, torch.zeros_like(self.f.cnt)]
        self.all_cnts = [self.f.all_cnt, torch.zeros_like(self.f.all_cnt)]
        self.host_cnts = [torch.zeros((self.world,), dtype=torch.int32, pin_memory=cuda) for _ in range(2)]
        self.events = [torch.cuda.Event() if cuda else None for _ in range(2)]
        self.frame = 0
        self.pending = None          # slot of the frame measured but not yet gathered / applied
        self.inflight = None         # (slot, gathered view, work) of the payload collective started by begin()

    def _start_payload(self, mapper, slot):
        if self.events[slot] is not None:
            self.events[slot].synchronize()          # recorded a frame ago: the counts are on the host by now
        n, used = self.f.sized(self.host_cnts[slot].tolist())
        self.f.used_records = used; self.f.used_bytes_total += used * self.f.BLOCK_BYTES
        self.f._order(mapper, "mapper", "torch")
        gathered, work = self.f.gather_payload(self.bufs[slot], n, async_op=True)
        self.inflight = (slot, gathered, work)

    def begin(self, mapper, depth, T_L_C, cam):
        if self.pending is not None:
            self._start_payload(mapper, self.pending); self.pending = None
        slot = self.frame & 1
        mapper.measure_depth(depth, T_L_C, cam, self.bufs[slot], self.cnts[slot])
        self.f._order(mapper, "mapper", "torch")
        self.f.gather_counts(self.cnts[slot], self.all_cnts[slot])
        self.host_cnts[slot].copy_(self.all_cnts[slot], non_blocking=True)
        if self.events[slot] is not None:
            self.events[slot].record(torch.cuda.current_stream(self.f.buf.device))
        self.pending = slot
        self.frame += 1

    def finish_previous(self, mapper):
        if self.inflight is None:
            return False
        slot, gathered, work = self.inflight; self.inflight = None
        if work is not None:
            work.wait()
        self.f._order(mapper, "torch", "mapper")
        self.f.apply(mapper, gathered, self.all_cnts[slot])
        return True

    def drain(self, mapper):
        """Apply what is still on its way; returns the number of frames applied (0, 1 or 2) -- the caller owes each its colour / ESDF."""
        n = 1 if self.finish_previous(mapper) else 0
        if self.pending is not None:
            self._start_payload(mapper, self.pending); self.pending = None
            n += 1 if self.finish_previous(mapper) else 0
        return n


def camera_yaw_offset_deg(rank, world):
    """Config 4 of BASELINE.json: cameras on the same rig circle at 45 degree yaw offsets."""
    return 45.0 * (rank % 8)
