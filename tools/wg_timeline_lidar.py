#!/usr/bin/env python3
"""Per-workgroup profile of k_mark_view<LidarSensor> (configs[4] scan) from the -DNVBX_WG_TIMES variant: start / end of every tile bundle,
time spent inside its flushes, number of flushes, keys sent to HBM.   NVBX_LIB=.../libnvblox_hip_wgt.so python tools/wg_timeline_lidar.py"""
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
for s in range(6):
    g.integrate_prepared(la[s % 4]); g.synchronize(); torch.cuda.synchronize(dev)
    fn(buf.ctypes.data_as(C.c_void_p), buf.size)
    b = buf[0].astype(np.int64); used = b[:, 0] > 0
    t0 = b[used, 0].min()
    st = (b[used, 0] - t0) / 100.0; en = (b[used, 7] - t0) / 100.0
    nfl = b[:, 6] >> 32; nkeys = b[:, 6] & 0xFFFFFFFF
    res.append({"workgroups": int(used.sum()), "launch_end_us": round(float(en.max()), 1), "start_median_us": round(float(np.median(st)), 1), "start_max_us": round(float(st.max()), 1),
                "dur_median_us": round(float(np.median(en - st)), 1), "dur_max_us": round(float((en - st).max()), 1),
                "flush_time_median_us": round(float(np.median(b[used, 2] / 100.0)), 1), "flush_time_sum_over_dur": round(float((b[used, 2] / 100.0).sum() / (en - st).sum()), 3),
                "flushes_median": float(np.median(nfl[used])), "keys_median": float(np.median(nkeys[used])), "keys_total": int(nkeys[used].sum())})
idx = np.nonzero(used)[0]; dur = en - st
order = np.argsort(-dur)[:12]
slow = [{"wg": int(idx[j]), "start": round(float(st[j]), 1), "dur": round(float(dur[j]), 1), "flush_time": round(float(b[idx[j], 2] / 100.0), 1),
         "flushes": int(nfl[idx[j]]), "keys": int(nkeys[idx[j]]),
         "last_flush_phases_us": {"compact": round(float((b[idx[j], 3] - b[idx[j], 1]) / 100.0), 2), "probe+claim": round(float((b[idx[j], 4] - b[idx[j], 3]) / 100.0), 2),
                                  "append": round(float((b[idx[j], 5] - b[idx[j], 4]) / 100.0), 2), "end": round(float((b[idx[j], 7] - b[idx[j], 5]) / 100.0), 2)}} for j in order]
hist = np.histogram(dur, bins=[0, 5, 10, 20, 30, 40, 60, 80, 200])[0].tolist()
late = [{"wg": int(idx[j]), "start": round(float(st[j]), 1), "dur": round(float(dur[j]), 1)} for j in np.argsort(-en)[:8]]
print(json.dumps({"duration_histogram_us[0,5,10,20,30,40,60,80,200]": hist, "slowest": slow, "last_to_end": late}))
print(json.dumps({"blocks_in_view": g.counters()["tsdf_blocks_in_view"], "samples": res}, indent=1))
