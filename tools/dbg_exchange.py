import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import helpers as H
from isaac_ros_nvblox_amd import mapper as M
cam = H.SMALL_CAM
n, ranks, warm = 9, 2, 5
fr = [H.frames(n, cam, color=True, stride=11), H.frames(n, cam, color=True, stride=11, yaw_offset_deg=180.0)]
dev = torch.device("cuda", 0)
def protocol(mode):
    ms = [M.Mapper(M.default_params(), block_capacity=1 << 14) for _ in range(ranks)]
    for m_ in ms:
        if mode == "classic": m_.set_color_deferral(False)
        if mode == "zc": m_.set_color_deferral(True, staged=False)
        for q in range(ranks):
            for d, rgb, T in fr[q]:
                m_.integrate_depth(d, T, cam)
        m_.update_esdf(); m_.synchronize()
    bufs = [[torch.zeros((4097, 3), dtype=torch.int32, device=dev) for _ in range(3)] for _ in range(ranks)]
    alls = [[torch.zeros((ranks, 4097, 3), dtype=torch.int32, device=dev) for _ in range(3)] for _ in range(ranks)]
    pending = [None] * ranks
    keep = []
    for i in range(warm + n):
        u, slot = i % n, i % 3
        for r in range(ranks):
            ms[r].set_view_export(bufs[r][slot]); ms[r].integrate_depth(fr[r][u][0], fr[r][u][2], cam)
        for r in range(ranks):
            ms[r].synchronize()
        for r in range(ranks):
            for q in range(ranks):
                alls[r][slot][q].copy_(bufs[q][slot])
        torch.cuda.synchronize(dev)
        for r in range(ranks):
            if pending[r] is not None:
                ms[r].mark_esdf_dirty_gathered(alls[r][pending[r]], ranks, r, 4096, deferred=True)
            pending[r] = slot
            ms[r].integrate_color(fr[r][u][1], fr[r][u][2], cam); ms[r].update_esdf()
    for r in range(ranks):
        ms[r].mark_esdf_dirty_gathered(alls[r][pending[r]], ranks, r, 4096)
        ms[r].set_view_export(None); ms[r].update_esdf(); ms[r].synchronize()
    return ms
def plain(r):
    g = M.Mapper(M.default_params(), block_capacity=1 << 14); g.set_color_deferral(False)
    for q in range(ranks):
        for d, rgb, T in fr[q]:
            g.integrate_depth(d, T, cam)
    g.update_esdf()
    for i in range(warm + n):
        u = i % n
        g.integrate_depth(fr[r][u][0], fr[r][u][2], cam); g.integrate_color(fr[r][u][1], fr[r][u][2], cam); g.update_esdf()
    g.update_esdf(); g.synchronize(); return g
def cset(g):
    return set(map(tuple, g.block_indices(M.LAYER_COLOR).tolist()))
ref = [cset(plain(r)) for r in range(ranks)]
print("plain colour blocks", [len(x) for x in ref])
for mode in ("staged", "staged", "staged", "zc", "classic"):
    ms = protocol(mode)
    for r in range(ranks):
        s = cset(ms[r])
        print(mode, "rank", r, len(s), "missing", sorted(ref[r] - s)[:5], "extra", sorted(s - ref[r])[:5])
