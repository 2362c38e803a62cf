"""One-camera-per-GPU sharding: all-gather of updated block indices before the ESDF sweep.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU in the tests).
The exchange is two fixed-size all-gathers (counts, then indices padded to `max_blocks`), so the message size is
static (<= 8 x 48 KiB at max_blocks = 4096): latency-bound on the point-to-point xGMI links, no host copy, no sync.
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
        self.idx = torch.zeros((self.max_blocks, 3), dtype=torch.int32, device=device)
        self.cnt = torch.zeros((1,), dtype=torch.int32, device=device)
        self.all_idx = torch.zeros((self.world, self.max_blocks, 3), dtype=torch.int32, device=device)
        self.all_cnt = torch.zeros((self.world,), dtype=torch.int32, device=device)

    def all_gather(self):
        """all-gather self.idx / self.cnt into self.all_idx / self.all_cnt (no-op copy when world == 1)."""
        if self.world == 1:
            self.all_idx[0].copy_(self.idx); self.all_cnt.copy_(self.cnt)
            return
        dist.all_gather_into_tensor(self.all_cnt, self.cnt, group=self.group)
        dist.all_gather_into_tensor(self.all_idx.view(-1, 3), self.idx, group=self.group)

    def exchange(self, mapper):
        """Export this GPU's dirty TSDF block indices, all-gather, mark every peer's blocks ESDF-dirty locally."""
        mapper.esdf_dirty_list(self.idx, self.cnt)
        self.all_gather()
        for r in range(self.world):
            if r != self.rank:
                mapper.mark_esdf_dirty(self.all_idx[r], self.all_cnt[r:r + 1], self.max_blocks)

    def union_host(self):
        """Host-side union of the gathered lists (tests / diagnostics only; synchronises)."""
        cnt = self.all_cnt.cpu().tolist(); idx = self.all_idx.cpu()
        out = set()
        for r in range(self.world):
            out |= set(map(tuple, idx[r, :min(cnt[r], self.max_blocks)].tolist()))
        return out


def camera_yaw_offset_deg(rank, world):
    """Config 4 of BASELINE.json: cameras on the same rig circle at 45 degree yaw offsets."""
    return 45.0 * (rank % 8)
