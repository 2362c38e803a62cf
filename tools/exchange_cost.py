"""GPU-side cost of the multi-GPU exchange launches on ONE GPU (no collective): the peer's list is a copy of the own list (worst case:
every block exists locally).  Usage: python tools/exchange_cost.py"""
import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from isaac_ros_nvblox_amd import mapper as M, synthetic as S
from isaac_ros_nvblox_amd.dist import DirtyBlockExchange
dev = torch.device("cuda", 0); cam = S.REPLICA_LIKE_CAM; sc = S.Scene()
fr = []
for i in range(50):
    T = S.trajectory_pose(i * 4, 200); d, rgb = S.render(sc, T, cam); fr.append((torch.from_numpy(d).to(dev), torch.from_numpy(rgb).to(dev), T))
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
g = M.Mapper(M.default_params(), device=0, block_capacity=1 << 15, stream=stream.cuda_stream)
da = [g.prepare_depth(d, T, cam) for d, _, T in fr]; ca = [g.prepare_color(c, T, cam) for _, c, T in fr]
ex = DirtyBlockExchange(4096, dev)
gathered = torch.zeros((2, 4097, 3), dtype=torch.int32, device=dev)
def step(i, mode):  # 0: single GPU; 1: + export launch; 2: + union step as its own launch; 3: union step riding in the colour launch; 4: 3 with the export written by the depth pass
    k = i % 50
    g.integrate_prepared(da[k])
    if 1 <= mode <= 3: g.esdf_dirty_list(ex.idx, ex.cnt)
    if mode >= 3: g.mark_esdf_dirty_gathered(gathered, 2, 0, 4096, deferred=True)
    g.integrate_prepared(ca[k])
    if mode == 2: g.mark_esdf_dirty_gathered(gathered, 2, 0, 4096)
    g.update_esdf()
for mode in (0, 1, 2, 3, 4):
    g.set_view_export(ex.buf if mode == 4 else None)
    for i in range(30): step(i, min(mode, 1))
    gathered[1].copy_(ex.buf)      # stands in for the all-gather result (the peer's list = an own list: every block exists locally)
    g.synchronize(); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(400): step(30 + i, mode)
    g.synchronize(); torch.cuda.synchronize()
    print("mode", mode, "ms/frame %.4f" % ((time.perf_counter() - t) / 400 * 1e3))
