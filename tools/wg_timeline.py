#!/usr/bin/env python3
"""Per-workgroup timeline of the two camera launches of the pipelined frame, from the -DNVBX_WG_TIMES variant of the library
(tools/build_variant.sh wgt "-DNVBX_WG_TIMES").  Usage on the GPU box:
    NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline.py [--samples 30] [--env K=V ...]
Time base: s_memrealtime (100 MHz, shared by all CUs), relative to the first workgroup's start of the launch; microseconds."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=30)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--classic", action="store_true", help="classic launch order: the view-marking launch carries the tiles and the held-back distance transform only")
    ap.add_argument("--window-us", type=float, default=12.0, help="a workgroup belongs to the last launch if it started within this many us of the launch's last start (the hall's launches run 17-25 us: 22)")
    ap.add_argument("--scene", default="room", choices=["room", "hall"], help="hall: bench.py --scene hall (14 x 12 x 3 m, ~2 400 blocks in view)")
    ap.add_argument("--tiles", type=int, default=88, help="tile workgroups of the view-marking launch (640x480, factor 4: 11 x 8 groups of 2 x 2 tiles)")
    args = ap.parse_args()
    import torch
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S, _lib
    lib = _lib.load()
    fn = lib.nvbx_debug_wg_times
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int64]
    max_wg = fn(None, 0)
    assert max_wg > 0
    cam = S.REPLICA_LIKE_CAM
    dev = torch.device("cuda", 0)
    sc = S.Scene() if args.scene == "room" else S.Scene(room_min=(-7.0, -6.0, 0.0), room_max=(7.0, 6.0, 3.0))
    from concurrent.futures import ThreadPoolExecutor
    def one(i):
        T = S.trajectory_pose(i * (200 // args.frames), 200)
        d, rgb = S.render(sc, T, cam)
        return d, rgb, T
    with ThreadPoolExecutor(16) as pool:
        fr = list(pool.map(one, range(args.frames)))
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    g = M.Mapper(M.default_params(), device=0, block_capacity=(1 << 14) if args.scene == "room" else (1 << 15), stream=stream.cuda_stream)
    g.set_color_deferral(not args.classic)
    da = [g.prepare_depth(torch.from_numpy(d).to(dev), T, cam) for d, _, T in fr]
    ca = [g.prepare_color(torch.from_numpy(c).to(dev), T, cam) for _, c, T in fr]
    for k in range(args.frames):      # build the map once (revisit steady state below)
        g.integrate_prepared(da[k]); g.integrate_prepared(ca[k]); g.update_esdf()
    g.synchronize()
    buf = np.zeros((2, max_wg, 8), np.uint64)
    acc = {0: [], 1: []}
    k = 0
    for s in range(args.samples):
        for _ in range(7 + s % 5):
            g.integrate_prepared(da[k % args.frames]); g.integrate_prepared(ca[k % args.frames]); g.update_esdf(); k += 1
        g.synchronize(); torch.cuda.synchronize(dev)
        fn(buf.ctypes.data_as(C.c_void_p), buf.size)
        for kern in (0, 1):
            b = buf[kern].astype(np.int64)
            used = b[:, 0] > 0
            if used.sum() == 0:
                continue
            used &= b[:, 0] > b[:, 0].max() - int(args.window_us * 100)               # the LAST launch only (grids differ from frame to frame: higher workgroups keep older stamps)
            t0 = b[used, 0].min()
            rel = np.where((b > 0) & used[:, None], (b - t0) / 100.0, np.nan)        # us
            rel[:, 6] = np.where(used, np.where(b[:, 6] > 10 ** 9, (b[:, 6] - t0) / 100.0, b[:, 6]), np.nan)          # slot 6 carries a count (or, in an experiment build, a time)
            acc[kern].append(rel[: int(np.nonzero(used)[0].max()) + 1])
    hw = g.counters()["blocks_allocated"]
    n_scan = min(256, 8 * ((hw + hw // 4 + 64 + 2047) // 2048))
    roles0 = [("tiles", args.tiles), ("trace", 600), ("scan", n_scan), ("mark", 256)]
    if args.classic:
        roles0 = [("edt", 256), ("tiles", args.tiles)]
    out = {}
    def summarize(samples, roles, name):
        n = min(x.shape[0] for x in samples)
        grids = sorted(set(x.shape[0] for x in samples))
        res = {"grid": grids}
        o = 0
        for role, cnt in roles:
            cnt = min(cnt, n - o)
            if cnt <= 0:
                break
            st = np.array([np.nanmin(x[o:o + cnt, 0]) for x in samples]); stl = np.array([np.nanmax(x[o:o + cnt, 0]) for x in samples])
            en_med = np.array([np.nanmedian(x[o:o + cnt, 7]) for x in samples]); en_max = np.array([np.nanmax(x[o:o + cnt, 7]) for x in samples])
            dur = np.array([np.nanmedian(x[o:o + cnt, 7] - x[o:o + cnt, 0]) for x in samples]); durmax = np.array([np.nanmax(x[o:o + cnt, 7] - x[o:o + cnt, 0]) for x in samples])
            r = {"n": int(cnt), "first_start": round(float(np.median(st)), 2), "last_start": round(float(np.median(stl)), 2), "end_median": round(float(np.median(en_med)), 2),
                 "end_max": round(float(np.median(en_max)), 2), "dur_median": round(float(np.median(dur)), 2), "dur_max": round(float(np.median(durmax)), 2)}
            if role == "tiles" and not args.classic or role == "tiles":
                ph = {}
                for i, nm in ((1, "depth+init"), (2, "walk"), (3, "compact"), (4, "probe+claim"), (5, "append")):
                    v = np.array([np.nanmedian(x[o:o + cnt, i] - x[o:o + cnt, 0]) for x in samples])
                    ph[nm] = round(float(np.nanmedian(v)), 2)
                r["phase_end_since_wg_start_median"] = ph
                r["keys_flushed_median_max"] = [float(np.median([np.nanmedian(x[o:o + cnt, 6]) for x in samples])), float(np.median([np.nanmax(x[o:o + cnt, 6]) for x in samples]))]
            if role == "trace":
                allr = np.concatenate([x[o:o + cnt, 6] for x in samples]); allr = allr[~np.isnan(allr)]
                r["rounds_hist"] = {str(int(v)): int(c) for v, c in zip(*np.unique(allr, return_counts=True))}
                alld = np.concatenate([(x[o:o + cnt, 7] - x[o:o + cnt, 0]) for x in samples]); 
                r["dur_by_rounds"] = {str(int(v)): round(float(np.nanmedian(alld[np.concatenate([x[o:o + cnt, 6] for x in samples]) == v])), 2) for v in np.unique(allr)}
            if role == "mark":
                alln = np.concatenate([x[o:o + cnt, 6] for x in samples]); alld = np.concatenate([(x[o:o + cnt, 7] - x[o:o + cnt, 0]) for x in samples])
                first = np.concatenate([(x[o:o + cnt, 1] - x[o:o + cnt, 0]) for x in samples]); loop_end = np.concatenate([(x[o:o + cnt, 2] - x[o:o + cnt, 0]) for x in samples])
                r["by_entries"] = {str(int(v)): {"workers": int((alln == v).sum()), "dur_median": round(float(np.nanmedian(alld[alln == v])), 2), "dur_max": round(float(np.nanmax(alld[alln == v])), 2),
                                                 "first_entry_done_median": (round(float(np.nanmedian(first[alln == v])), 2) if v > 0 else None),
                                                 "loop_end_median": round(float(np.nanmedian(loop_end[alln == v])), 2)} for v in np.unique(alln[~np.isnan(alln)])}
            # the slowest workgroups of the role in the last few samples: every stamp they left (relative to the launch's first start)
            slow = []
            for x in samples[-6:]:
                seg = x[o:o + cnt]
                for j in np.argsort(-np.nan_to_num(seg[:, 7]))[:2]:
                    slow.append({"wg": int(o + j), "stamps": [None if np.isnan(v) else round(float(v), 2) for v in seg[j]]})
            r["slowest"] = slow
            res[role] = r
            o += cnt
        res["launch_end"] = round(float(np.median([np.nanmax(x[:, 7]) for x in samples])), 2)
        out[name] = res
    if acc[0]:
        summarize(acc[0], roles0, "k_mark_view")
    if acc[1]:
        # roles of the fused launch by the role-specific stamp each worker leaves (slot 1 = distance transform, 2 = TSDF update, 3 = colour)
        res = {"grid": sorted(set(x.shape[0] for x in acc[1]))}
        for role, slot in (("edt", 1), ("tsdf", 2), ("colour", 3)):
            st, en, enmax, dur, durmax, cnt = [], [], [], [], [], []
            for x in acc[1]:
                sel = ~np.isnan(x[:, slot])
                if sel.sum() == 0:
                    continue
                st.append(np.nanmin(x[sel, 0])); en.append(np.nanmedian(x[sel, 7])); enmax.append(np.nanmax(x[sel, 7]))
                dur.append(np.nanmedian(x[sel, 7] - x[sel, 0])); durmax.append(np.nanmax(x[sel, 7] - x[sel, 0])); cnt.append(int(sel.sum()))
            if cnt:
                res[role] = {"n": int(np.median(cnt)), "first_start": round(float(np.median(st)), 2), "end_median": round(float(np.median(en)), 2),
                             "end_max": round(float(np.median(enmax)), 2), "dur_median": round(float(np.median(dur)), 2), "dur_max": round(float(np.median(durmax)), 2)}
        res["launch_end"] = round(float(np.median([np.nanmax(x[:, 7]) for x in acc[1]])), 2)
        out["k_integrate_tsdf_color"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
