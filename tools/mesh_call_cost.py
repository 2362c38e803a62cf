#!/usr/bin/env python3
"""Where the time of a per-frame updateColorMesh goes (34 us per call against a 13 us kernel): wall-clock loops of
depth / depth + mesh / depth + colour + esdf / ... + mesh on the camera workload, classic and pipelined order, with the launches each loop makes."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from isaac_ros_nvblox_amd import mapper as M, synthetic as S
import bench

cam = S.REPLICA_LIKE_CAM; dev = torch.device("cuda", 0)
sc = S.Scene(); nu = 40
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as pool:
    fr = list(pool.map(lambda i: (S.render(sc, S.trajectory_pose(i * 5, 200), cam), S.trajectory_pose(i * 5, 200)), range(nu)))
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
g = M.Mapper(M.default_params(), device=0, block_capacity=1 << 14, stream=stream.cuda_stream)
da = [g.prepare_depth(torch.from_numpy(d).to(dev), T, cam) for (d, c), T in fr]
ca = [g.prepare_color(torch.from_numpy(c).to(dev), T, cam) for (d, c), T in fr]
out = {}
for deferral in (False, True):
    g.set_color_deferral(deferral)
    def loop(name, body, n=400):
        for i in range(40): body(i)
        g.synchronize(); torch.cuda.synchronize(dev)
        g.set_profiling(True)
        t = time.perf_counter()
        for i in range(n): body(i)
        g.synchronize(); torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / n * 1e6
        prof = g.profile(); g.set_profiling(False); prof.pop("_empty_event_pair", None)
        out["%s%s" % (name, " (deferral)" if deferral else "")] = {"wall_us_per_step": round(dt, 2), "launches_per_step": round(sum(v["count"] for v in prof.values()) / n, 2),
                  "kernel_span_us_per_step": round(sum(v["total_ms"] for v in prof.values()) / n * 1e3, 2),
                  "kernels_us": {bench.short(k): round(v["total_ms"] / v["count"] * 1e3, 2) for k, v in prof.items()}}
    loop("depth", lambda i: g.integrate_prepared(da[i % nu]))
    loop("depth+mesh", lambda i: (g.integrate_prepared(da[i % nu]), g.update_color_mesh()))
    loop("depth+colour+esdf", lambda i: (g.integrate_prepared(da[i % nu]), g.integrate_prepared(ca[i % nu]), g.update_esdf()))
    loop("depth+colour+esdf+mesh", lambda i: (g.integrate_prepared(da[i % nu]), g.integrate_prepared(ca[i % nu]), g.update_esdf(), g.update_color_mesh()))
    loop("depth+colour+esdf, mesh every 8th", lambda i: (g.integrate_prepared(da[i % nu]), g.integrate_prepared(ca[i % nu]), g.update_esdf(), g.update_color_mesh() if i % 8 == 7 else None))
print(json.dumps(out, indent=1))
