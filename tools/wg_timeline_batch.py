#!/usr/bin/env python3
"""Per-workgroup timeline of the two fused launches of a CAMERA BATCH step (bench.py --workload multicam), from the -DNVBX_WG_TIMES variant:
start / end of every workgroup by index range (riders first, then the tiles: k_mark_view<..., 8>; distance transform, TSDF update, colour: k_integrate_tsdf_color<..., 8>).
    NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline_batch.py [--cameras 8]"""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cameras", type=int, default=8); ap.add_argument("--samples", type=int, default=12)
    a = ap.parse_args()
    import torch
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S, _lib
    from isaac_ros_nvblox_amd.dist import camera_yaw_offset_deg
    lib = _lib.load(); fn = lib.nvbx_debug_wg_times; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int64]
    max_wg = fn(None, 0)
    cam = S.REPLICA_LIKE_CAM; rows, cols = cam[5], cam[4]; dev = torch.device("cuda", 0); sc = S.Scene(); nf = 24; nc = a.cameras
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    g = M.Mapper(M.default_params(), device=0, block_capacity=1 << 15, stream=stream.cuda_stream)
    fr = []
    for i in range(nf):
        per = []
        for c in range(nc):
            T = S.trajectory_pose(i * 8, 200, yaw_offset_deg=camera_yaw_offset_deg(c, nc)); d, rgb = S.render(sc, T, cam); per.append((d, rgb, T))
        fr.append(per)
    dd = [[torch.from_numpy(d).to(dev) for d, _, _ in per] for per in fr]
    cc = [[M.ColorFrame(rows, cols, 3, 0).write(torch.from_numpy(c).to(dev), stream.cuda_stream) for _, c, _ in per] for per in fr]
    da = [g.prepare_depth_batch(dd[i], [T for _, _, T in fr[i]], cam) for i in range(nf)]
    ca = [g.prepare_color_batch(cc[i], [T for _, _, T in fr[i]], cam) for i in range(nf)]

    def step(i):
        g.integrate_prepared_batch(da[i % nf]); g.integrate_prepared_batch(ca[i % nf]); g.update_esdf()
    for i in range(2 * nf):
        step(i)
    g.synchronize()
    buf = np.zeros((2, max_wg, 8), np.uint64); k = 0; res = {0: [], 1: []}
    for s in range(a.samples):
        for _ in range(5 + s % 3):
            step(k); k += 1
        g.synchronize(); torch.cuda.synchronize(dev)
        fn(buf.ctypes.data_as(C.c_void_p), buf.size)
        for kern in (0, 1):
            b = buf[kern].astype(np.int64); used = (b[:, 0] > 0) & (b[:, 7] > 0)
            if not used.any():
                continue
            used &= b[:, 0] > b[:, 0].max() - 3000
            t0 = b[used, 0].min(); idx = np.nonzero(used)[0]
            res[kern].append((idx, (b[idx, 0] - t0) / 100.0, (b[idx, 7] - t0) / 100.0))
    out = {}
    for kern, name in ((0, "k_mark_view"), (1, "k_integrate_tsdf_color")):
        if not res[kern]:
            continue
        idx, st, en = res[kern][-1]
        n = int(idx.max()) + 1; chunks = []
        for lo in range(0, n, max(1, n // 24)):
            hi = min(n, lo + max(1, n // 24)); sel = (idx >= lo) & (idx < hi)
            if sel.any():
                chunks.append({"wg": [lo, hi], "n": int(sel.sum()), "start_med": round(float(np.median(st[sel])), 1), "end_med": round(float(np.median(en[sel])), 1),
                               "end_max": round(float(en[sel].max()), 1), "dur_med": round(float(np.median(en[sel] - st[sel])), 1)})
        out[name] = {"workgroups": n, "launch_end_us_median_over_samples": round(float(np.median([e.max() for _, _, e in res[kern]])), 1), "by_index_range": chunks}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
