"""Per-kernel times of the camera frame with parts of it switched off: which launches are long because of what RIDES in them
(the held-back distance transform in k_mark_view, the ESDF site marking in k_integrate_color) and which by themselves.
Usage (GPU box): python tools/kernel_isolation.py   -> one line per mode"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_ros_nvblox_amd import mapper as M, synthetic as S

dev = torch.device("cuda", 0)
cam = S.REPLICA_LIKE_CAM
sc = S.Scene()
frames = []
for i in range(40):
    T = S.trajectory_pose(i, 400)
    d, rgb = S.render(sc, T, cam, color=True)
    frames.append((torch.from_numpy(d).to(dev), torch.from_numpy(rgb).to(dev), T))
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
for mode in ("depth", "depth+colour", "depth+esdf", "depth+colour+esdf"):
    g = M.Mapper(M.default_params(), device=0, block_capacity=1 << 14, stream=stream.cuda_stream)
    def step(k):
        d, c, T = frames[k % len(frames)]
        g.integrate_depth(d, T, cam)
        if "colour" in mode: g.integrate_color(c, T, cam)
        if "esdf" in mode: g.update_esdf()
    for k in range(40): step(k)
    g.synchronize(); g.set_profiling(True)
    for k in range(40, 240): step(k)
    g.synchronize()
    prof = g.profile(); g.set_profiling(False)
    print(mode.ljust(20), {k.strip().split("(")[0].split("<")[0].replace("void ", ""): round(v["total_ms"] / v["count"] * 1e3, 1) for k, v in prof.items() if not k.startswith("_")})
