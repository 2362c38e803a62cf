#!/usr/bin/env python3
"""Per-component ms/frame of the CPU oracle (the 'CPU reference' of BASELINE.md) on this host, for the thread count in
OMP_NUM_THREADS.  Same synthetic sequence and parameters as bench.py.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from isaac_ros_nvblox_amd import synthetic as S  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    cam = S.REPLICA_LIKE_CAM
    sc = S.Scene()
    fr = []
    for i in range(n + 2):
        T = S.trajectory_pose(i * 4, 200); d, rgb = S.render(sc, T, cam); fr.append((d, rgb, T))
    o = oracle.OracleMap(oracle.default_params())
    for d, rgb, T in fr[:2]:
        o.integrate_depth(d, T, cam); o.integrate_color(rgb, T, cam); o.update_esdf(); o.update_mesh()
    acc = dict(tsdf=0.0, color=0.0, esdf=0.0, mesh=0.0)
    for d, rgb, T in fr[2:]:
        t = time.perf_counter(); o.integrate_depth(d, T, cam); acc["tsdf"] += time.perf_counter() - t
        t = time.perf_counter(); o.integrate_color(rgb, T, cam); acc["color"] += time.perf_counter() - t
        t = time.perf_counter(); o.update_esdf(); acc["esdf"] += time.perf_counter() - t
        t = time.perf_counter(); o.update_mesh(); acc["mesh"] += time.perf_counter() - t
    out = {k: round(v / n * 1e3, 3) for k, v in acc.items()}
    out["tsdf_color_esdf"] = round(out["tsdf"] + out["color"] + out["esdf"], 3)
    out["threads"] = oracle.num_threads(); out["frames"] = n
    print(json.dumps(out))


if __name__ == "__main__":
    main()
