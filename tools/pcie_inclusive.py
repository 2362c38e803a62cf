#!/usr/bin/env python3
"""The camera step (integrateDepth + integrateColor + updateEsdf, 640x480, fuser.yaml parameters) when the boundary is handed HOST buffers: every frame's
depth (f32, 1.2 MB) and colour (rgb8, 0.9 MB) image is uploaded from pinned host memory first (what nvblox::Image::copyFromAsync does in the facade).
Never bench.py's `value` (that one has its inputs resident in HBM); DESIGN.md 5 quotes this beside it.

  same_stream : the uploads are enqueued on the mapper's stream in front of the calls (a host that only swaps the library: the reference node converts and
                copies on the mapper's stream, nvblox_node.cpp:1237-1263)
  copy_stream : the uploads run on a stream of their own, one frame ahead, three rotating depth buffers, events both ways; colour goes through a library
                frame (the mapper retains it, the writer continues in another one)
  resident    : the same loop without uploads

  python tools/pcie_inclusive.py > gpurun_out/pcie_inclusive.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from isaac_ros_nvblox_amd import mapper as M, synthetic as S

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
cam = S.REPLICA_LIKE_CAM; rows, cols = int(cam[5]), int(cam[4])
NU, STEPS, BLOCKS = 40, 200, 4
scene = S.Scene()
from concurrent.futures import ThreadPoolExecutor


def one(i):
    T = S.trajectory_pose(i * (200 // NU), 200)
    d, rgb = S.render(scene, T, cam)
    return d, rgb, T


with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
    host = list(pool.map(one, range(NU)))
depth_h = [torch.from_numpy(np.ascontiguousarray(d, np.float32)).pin_memory() for d, _, _ in host]
rgb_h = [torch.from_numpy(np.ascontiguousarray(c, np.uint8)).pin_memory() for _, c, _ in host]
poses = [T for _, _, T in host]
stream = torch.cuda.Stream(dev); copy_stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
g = M.Mapper(M.default_params(), device=0, block_capacity=1 << 14, stream=stream.cuda_stream)
dbuf = [torch.empty((rows, cols), dtype=torch.float32, device=dev) for _ in range(3)]
dargs = [[g.prepare_depth(dbuf[j], poses[k], cam) for k in range(NU)] for j in range(3)]
# the same with the depth image as uint16 millimetres (the encoding a depth camera delivers; nvbx_integrate_depth_u16mm converts in the kernels): half the bytes
depth16_h = [torch.from_numpy(np.round(np.clip(np.asarray(d, np.float64) * 1000.0, 0, 65535)).astype(np.uint16).view(np.int16)).pin_memory() for d, _, _ in host]
dbuf16 = [torch.empty((rows, cols), dtype=torch.int16, device=dev) for _ in range(3)]
Tn = [M.Mapper._T(T) for T in poses]; kc = M.Mapper._cam(cam)
import ctypes as C
frame = M.ColorFrame(rows, cols, 3, 0)
# resident copies for the comparison loop
depth_d = [t.to(dev) for t in depth_h]
rgb_f = [M.ColorFrame(rows, cols, 3, 0).write(t.to(dev), stream.cuda_stream) for t in rgb_h]
dargs_res = [g.prepare_depth(depth_d[k], poses[k], cam) for k in range(NU)]
cargs_res = [g.prepare_color(rgb_f[k], poses[k], cam) for k in range(NU)]


def step_resident(i):
    k = i % NU
    g.integrate_prepared(dargs_res[k]); g.integrate_prepared(cargs_res[k]); g.update_esdf()


def step_same_stream(i):
    k = i % NU; j = i % 3
    dbuf[j].copy_(depth_h[k], non_blocking=True)                    # (torch's current stream IS the mapper's stream)
    g.integrate_prepared(dargs[j][k])
    frame.write(rgb_h[k], stream.cuda_stream)                          # continues in another frame while the mapper holds the last one
    g.integrate_color(frame, poses[k], cam); g.update_esdf()


ready = [torch.cuda.Event() for _ in range(3)]; free = [torch.cuda.Event() for _ in range(3)]; cready = torch.cuda.Event()
state = {"primed": -1}


def upload(i):
    k = i % NU; j = i % 3
    copy_stream.wait_event(free[j])
    with torch.cuda.stream(copy_stream):
        dbuf[j].copy_(depth_h[k], non_blocking=True)
    ready[j].record(copy_stream)


def step_copy_stream(i):
    k = i % NU; j = i % 3
    if state["primed"] < i:
        upload(i); state["primed"] = i
    stream.wait_event(ready[j])
    g.integrate_prepared(dargs[j][k])
    free[j].record(stream)
    upload(i + 1); state["primed"] = i + 1                            # the next frame's depth image, beside this frame's launches
    frame.write(rgb_h[k], copy_stream.cuda_stream); cready.record(copy_stream)
    stream.wait_event(cready)
    g.integrate_color(frame, poses[k], cam); g.update_esdf()


def upload16(i):
    k = i % NU; j = i % 3
    copy_stream.wait_event(free[j])
    with torch.cuda.stream(copy_stream):
        dbuf16[j].copy_(depth16_h[k], non_blocking=True)
    ready[j].record(copy_stream)


def step_copy_stream_u16(i):
    k = i % NU; j = i % 3
    if state["primed"] < i:
        upload16(i); state["primed"] = i
    stream.wait_event(ready[j])
    g._check(g.lib.nvbx_integrate_depth_u16mm(g._h, C.c_void_p(dbuf16[j].data_ptr()), rows, cols, Tn[k].ctypes.data_as(C.c_void_p), C.byref(kc)))
    free[j].record(stream)
    upload16(i + 1); state["primed"] = i + 1
    frame.write(rgb_h[k], copy_stream.cuda_stream); cready.record(copy_stream)
    stream.wait_event(cready)
    g.integrate_color(frame, poses[k], cam); g.update_esdf()


def timed(step):
    for j in range(3):
        free[j].record(stream)
    state["primed"] = -1
    g.clear()
    for i in range(NU):
        step(i)
    g.synchronize(); torch.cuda.synchronize(dev)
    out = []
    base = NU
    for _ in range(BLOCKS):
        t0 = time.perf_counter()
        for i in range(base, base + STEPS):
            step(i)
        g.synchronize(); torch.cuda.synchronize(dev)
        out.append((time.perf_counter() - t0) / STEPS * 1e3); base += STEPS
    return out


import gc
gc.collect(); gc.freeze(); gc.disable()
res = {}
for name, fn in (("resident", step_resident), ("same_stream", step_same_stream), ("copy_stream", step_copy_stream), ("copy_stream_depth_u16mm", step_copy_stream_u16), ("resident_again", step_resident)):
    v = timed(fn)
    res[name] = {"ms_per_frame_median": round(float(np.median(v)), 4), "blocks": [round(x, 4) for x in v]}
# the bare uploads, for scale
torch.cuda.synchronize(dev); t0 = time.perf_counter()
for i in range(400):
    dbuf[i % 3].copy_(depth_h[i % NU], non_blocking=True); frame.write(rgb_h[i % NU], stream.cuda_stream)
torch.cuda.synchronize(dev); up = (time.perf_counter() - t0) / 400
h2d = rows * cols * 4 + rows * cols * 3
res["upload_only"] = {"ms_per_frame": round(up * 1e3, 4), "GBps": round(h2d / up / 1e9, 2)}
res["h2d_bytes_per_frame"] = h2d; res["h2d_bytes_per_frame_depth_u16mm"] = rows * cols * 2 + rows * cols * 3
res["workload"] = "configs[1] camera step, %d distinct 640x480 frames on the allocated map, blocks of %d frames" % (NU, STEPS)
print(json.dumps(res, indent=1))
