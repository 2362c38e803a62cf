#!/usr/bin/env python3
"""Per-workgroup profile of k_mark_view_grid (configs[4] scan) from the -DNVBX_WG_TIMES variant: when each bundle of rays starts, when its depth
pixel has arrived, when its lanes stand at their segments, and the first chunk's walk / word loads / atomics / reservation.
  tools/build_variant.sh wgt "-DNVBX_WG_TIMES";  NVBX_LIB=isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline_lidar_grid.py"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from isaac_ros_nvblox_amd import mapper as M, synthetic as S, _lib
lib = _lib.load()
fn = lib.nvbx_debug_wg_times; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int64]
max_wg = fn(None, 0)
dev = torch.device("cuda", 0)
lidar = S.SPINNING_LIDAR; sc = S.LidarScene()
scans = []
for i in range(4):
    T = S.lidar_pose(i, 400); scans.append((torch.from_numpy(S.render_lidar(sc, T, lidar, max_range=200.0)).to(dev), T))
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
g = M.Mapper(M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2), device=0, block_capacity=1 << 19, stream=stream.cuda_stream)
la = [g.prepare_lidar(r, T, lidar) for r, T in scans]
for k in range(8):
    g.integrate_prepared(la[k % 4])
g.synchronize()
buf = np.zeros((2, max_wg, 8), np.uint64)
res = []
names = ["depth+setup", "jump", "walk", "near loads", "rest"]
for s in range(6):
    g.integrate_prepared(la[s % 4]); g.synchronize(); torch.cuda.synchronize(dev)
    fn(buf.ctypes.data_as(C.c_void_p), buf.size)
    b = buf[0].astype(np.int64); used = (b[:, 0] > 0) & (b[:, 7] > 0)
    t0 = b[used, 0].min()
    st = (b[used, 0] - t0) / 100.0; en = (b[used, 7] - t0) / 100.0
    ph = {}
    stamps = [0, 1, 2, 3, 4, 7]
    full = used & (b[:, 3] > 0) & (b[:, 4] > 0)
    for j, nm in enumerate(names):
        a_, b_ = stamps[j], stamps[j + 1]
        d = (b[full, b_] - b[full, a_]) / 100.0
        ph[nm] = {"median": round(float(np.median(d)), 2), "p90": round(float(np.percentile(d, 90)), 2), "max": round(float(d.max()), 2)}
    info = b[:, 6]
    res.append({"workgroups": int(used.sum()), "with_a_far_chunk": int(full.sum()), "launch_end_us": round(float(en.max()), 1), "start_median_us": round(float(np.median(st)), 1),
                "start_p90_us": round(float(np.percentile(st, 90)), 1), "start_max_us": round(float(st.max()), 1),
                "dur_median_us": round(float(np.median(en - st)), 1), "dur_p90_us": round(float(np.percentile(en - st, 90)), 1), "dur_max_us": round(float((en - st).max()), 1),
                "phases_us": ph})
print(json.dumps({"blocks_in_view": g.counters()["tsdf_blocks_in_view"], "samples": res[-2:]}, indent=1))
idx = np.nonzero(used)[0]; dur = en - st
def row(j):
    w = idx[j]; t = b[w]
    return {"wg": int(w), "start": round(float(st[j]), 1), "dur": round(float(dur[j]), 1),             "phases": [round(float((t[stamps[q + 1]] - t[stamps[q]]) / 100.0), 1) if t[stamps[q + 1]] > 0 and t[stamps[q]] > 0 else None for q in range(5)]}
print(json.dumps({"slowest": [row(j) for j in np.argsort(-dur)[:14]]}))
print(json.dumps({"last_to_start": [row(j) for j in np.argsort(-st)[:6]]}))
print(json.dumps({"start_hist_us[0,2,5,10,20,30,40,50,60,80]": np.histogram(st, bins=[0, 2, 5, 10, 20, 30, 40, 50, 60, 80])[0].tolist(),
                  "end_hist_us": np.histogram(en, bins=[0, 2, 5, 10, 20, 30, 40, 50, 60, 80])[0].tolist()}))
