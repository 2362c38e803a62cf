#!/usr/bin/env python3
"""bench.py -- ms/frame and frames/s of the TSDF + Color + ESDF hot path (mesh timed beside it) on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d [D]): synthetic Replica-like room, 640x480 depth + colour, 0.05 m voxels,
fuser.yaml integrator parameters, camera on the circle trajectory; ESDF updated every frame (worst case).  A "step" is
one frame: integrateDepth + integrateColor + updateEsdf, inputs already resident in HBM.  N > 1: one camera per GPU
(45 degree yaw offsets on the same rig, config 4), block-index all-gather over RCCL before every ESDF sweep, weak scaling.

Prints ONE JSON line (rank 0).  `value` = frames/s of the whole job (all ranks' frames / max-over-ranks time).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

README_RTX5090_MS = {"tsdf": 0.1, "color": 0.3, "esdf": 0.3, "mesh": 0.3}   # /root/reference README.md:69-97 (Replica)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(kernel, c, rows, cols):
    """SURVEY.md 8(d) per-frame algorithmic bytes of one launch of `kernel`, from the measured per-frame counts."""
    B = 4096
    Nv, Nc, Na = c["tsdf_blocks_in_view"], c["color_blocks_updated"], c["blocks_allocated"]
    Nu, Ne, Wv = c["esdf_columns_marked"], c["esdf_blocks_swept"], c["esdf_window_voxels"]
    if kernel.startswith("k_integrate_tsdf"):
        return rows * cols * 4 + Nv * (12 + 8) + Nv * B * 2
    if kernel.startswith("k_mark_view"):
        return (rows // 4) * (cols // 4) * 4 + Nv * 16 * 2          # sub-sampled depth read + one hash entry RMW per block in view
    if kernel.startswith("k_integrate_color"):
        # SURVEY 8(d) bytes_color minus the sphere tracer's share: colour image + synthetic depth read + TSDF of the band blocks
        # + colour RMW of the band blocks, plus the ESDF site marking that rides in the same launch (bytes_esdf's first term:
        # TSDF z-band of the re-marked columns read; slice plane masks written).  (The kernel itself no longer reads the TSDF for
        # the band vote -- an exact per-block flag kept by the TSDF writers decides it -- so its traffic is below this figure.)
        return rows * cols * 3 + (rows // 4) * (cols // 4) * 4 + Nc * B + Nc * B * 2 + Nu * 2 * B + Nu * 520
    if kernel.startswith("k_sphere_trace"):
        return (rows // 4) * (cols // 4) * 4 + Nc * B                # synthetic depth write + TSDF blocks read once
    if kernel.startswith("k_esdf_mark"):
        return Nu * 2 * B + Nu * 512 + Nu * 8                       # TSDF z-band (k_z = 2 blocks) read + slice plane + site mask written
    if kernel.startswith("k_esdf_edt"):
        return Ne * (512 * 2 + 121 * (16 + 8))                      # plane RMW + 11x11 neighbour hash entries and site masks per swept block
    if kernel.startswith("k_mesh"):
        return int(c["mesh_blocks_updated"] * B * (1.42 + 1.0) + c["mesh_vertices"] * 28 + c["mesh_triangles"] * 12)
    return 0


def main_lidar(args):
    """configs[4]: one step = integrateDepth of one 1024x64 LiDAR scan (range image resident in HBM), 0.10 m voxels, 200 m."""
    import torch
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    assert args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1, "the LiDAR workload line is single-GPU"
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    lidar = S.SPINNING_LIDAR
    sc = S.LidarScene()
    nu = max(2, min(args.unique_frames, 16))
    scans = []
    for i in range(nu):
        T = S.lidar_pose(i, 400)
        scans.append((torch.from_numpy(S.render_lidar(sc, T, lidar, max_range=200.0)).to(dev), T))
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    p = M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2)
    g = M.Mapper(p, device=0, block_capacity=1 << 19, stream=stream.cuda_stream)
    largs = [g.prepare_lidar(r, T, lidar) for r, T in scans]
    for i in range(args.warmup):
        g.integrate_prepared(largs[i % nu])
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    for i in range(args.steps):
        g.integrate_prepared(largs[(args.warmup + i) % nu])
    torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    g.set_profiling(True)
    nv = []
    for i in range(min(args.steps, 40)):
        g.integrate_prepared(largs[i % nu]); nv.append(g.counters()["tsdf_blocks_in_view"])
    prof = g.profile(); g.set_profiling(False)
    c = g.counters(); Nv = float(np.mean(nv))
    kern = {}
    prof.pop("_empty_event_pair", None)
    raw_sum_us = sum(v_["total_ms"] / v_["count"] * 1e3 for v_ in prof.values())
    ev_overhead_us = max(0.0, (raw_sum_us - ms * 1e3) / max(1, len(prof)))     # launches must add up to the un-instrumented scan time
    for k_, v_ in prof.items():
        us = max(0.1, v_["total_ms"] / v_["count"] * 1e3 - ev_overhead_us)
        name = "k_integrate_tsdf" if "integrate" in k_ else "k_mark_view"
        ab = (64 * 1024 * 4 + Nv * (16 + 4096 * 2)) if name == "k_integrate_tsdf" else (32 * 512 * 4 + Nv * 16 * 2)
        kern[name] = {"avg_us": round(us, 2), "algorithmic_bytes": int(ab), "achieved_GBps": round(ab / (us * 1e-6) / 1e9, 1)}
    dom = max(kern, key=lambda k_: kern[k_]["avg_us"])
    out = {"metric": "scans/s, LiDAR projective TSDF integrate, synthetic 1024x64 @0.10m, 200 m", "value": round(args.steps / dt, 2),
           "unit": "scans/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[4]: synthetic 1024x64 spinning LiDAR (SURVEY 8d), 0.10 m voxels, 200 m range, ray subsampling 2"},
           "per_scan_counts": {"tsdf_blocks_in_view": round(Nv, 1), "blocks_allocated": c["blocks_allocated"], "capacity_overflow": c["capacity_overflow"]},
           "kernels": kern,
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(kern[dom]["achieved_GBps"] / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_launch": kern[dom]["algorithmic_bytes"], "avg_launch_us": kern[dom]["avg_us"]},
           "cpu_baseline": None}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--unique-frames", type=int, default=50, help="distinct rendered frames cycled through (HBM-resident)")
    ap.add_argument("--cpu-frames", type=int, default=48, help="minimum number of frames of the same workload timed on the CPU oracle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU baseline keeps integrating (cycling the same frames) until this much CPU wall time has passed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="camera", choices=["camera", "lidar"],
                    help="camera = BASELINE.json configs[1] (the metric's configuration, default); lidar = configs[4] "
                         "(1024x64 spinning LiDAR, 0.10 m voxels, 200 m), the configuration where HBM bytes dominate")
    args = ap.parse_args()
    if args.workload == "lidar":
        return main_lidar(args)

    import torch
    import torch.distributed as dist
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    from isaac_ros_nvblox_amd.dist import PipelinedDirtyBlockExchange, camera_yaw_offset_deg

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NVBX_BENCH_SAME_DEVICE") == "1":
        local_rank = 0     # control-flow check of the N > 1 path on a 1-GPU box (with NVBX_BENCH_BACKEND=gloo); not a measurement
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("NVBX_BENCH_BACKEND", "nccl")      # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cam = S.REPLICA_LIKE_CAM
    rows, cols = cam[5], cam[4]
    scene = S.Scene()
    nu = max(2, min(args.unique_frames, args.steps + args.warmup))
    stride = max(1, 200 // nu)
    yaw = camera_yaw_offset_deg(rank, world)
    host_frames = []
    for i in range(nu):
        T = S.trajectory_pose(i * stride, 200, yaw_offset_deg=yaw)
        d, rgb = S.render(scene, T, cam)
        host_frames.append((d, rgb, T))
    depth_dev = [torch.from_numpy(d).to(dev) for d, _, _ in host_frames]
    rgb_dev = [torch.from_numpy(c).to(dev) for _, c, _ in host_frames]
    poses = [T for _, _, T in host_frames]

    stream = torch.cuda.Stream(dev)      # one explicit stream for torch ops, RCCL hand-off and every mapper kernel
    torch.cuda.set_stream(stream)
    g = M.Mapper(M.default_params(), device=local_rank, block_capacity=1 << 15, stream=stream.cuda_stream)
    ex = PipelinedDirtyBlockExchange(4096, dev) if world > 1 else None      # one packed all-gather per frame, joined one frame later

    dargs = [g.prepare_depth(depth_dev[k], poses[k], cam) for k in range(nu)]
    cargs = [g.prepare_color(rgb_dev[k], poses[k], cam) for k in range(nu)]

    def step(i, mesh=False, exchange=True):
        k = i % nu
        xg = ex if exchange else None            # rank-0-only passes after the timed region must not enter a collective
        if xg is not None:
            xg.before_depth(g)                   # the depth pass writes this frame's block indices into the exchange buffer itself
        g.integrate_prepared(dargs[k])          # MultiMapper::integrateDepth
        if xg is not None:
            xg.start(g)                          # dirty block indices: export + async RCCL all-gather (needs only the depth pass)
            xg.finish_previous(g, deferred=True) # join the PREVIOUS frame's all-gather; its union step rides in the colour launch below
        g.integrate_prepared(cargs[k])          # MultiMapper::integrateColor (+ marking of own and peers' dirty blocks)
        g.update_esdf()                          # MultiMapper::updateEsdf
        if mesh:
            g.update_color_mesh()

    def barrier():
        if ex is not None:
            ex.drain(g)          # the all-gather still in flight is joined and applied inside the timed region
            g.set_view_export(None)
        g.synchronize()          # launches anything the mapper holds back (the EDT of the last updateEsdf) and waits for its stream
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    fps = world * args.steps / dt

    if rank != 0:
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- rank 0 extras (outside the timed region): per-component times, per-kernel roofline, CPU baseline
    def timed(fn, n):
        g.synchronize(); torch.cuda.synchronize(dev); t = time.perf_counter()
        for i in range(n):
            fn(i)
        g.synchronize(); torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / n * 1e3

    base = args.warmup + args.steps
    n2 = min(args.steps, 100)
    comp = {}
    comp["tsdf"] = timed(lambda i: g.integrate_depth(depth_dev[(base + i) % nu], poses[(base + i) % nu], cam), n2)
    comp["color"] = timed(lambda i: g.integrate_color(rgb_dev[(base + i) % nu], poses[(base + i) % nu], cam), n2)

    def esdf_only(i):
        g.integrate_depth(depth_dev[(base + i) % nu], poses[(base + i) % nu], cam); g.update_esdf()
    comp["esdf"] = max(0.0, timed(esdf_only, n2) - comp["tsdf"])

    def mesh_only(i):
        g.integrate_depth(depth_dev[(base + i) % nu], poses[(base + i) % nu], cam); g.update_color_mesh()
    comp["mesh"] = max(0.0, timed(mesh_only, n2) - comp["tsdf"])

    # per-frame latency (SURVEY 8d timing protocol): every frame is waited for, so this is the latency a caller sees, not the
    # pipelined throughput of the timed region.  wall = host clock around the three calls + synchronize (includes the host's
    # wake-up after the stream drains); gpu = hipEvent pair on the mapper stream around the same three calls.
    lat_wall, lat_gpu = [], []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n2)]
    for i in range(n2):
        t = time.perf_counter()
        ev[i][0].record(stream)
        step(base + i, exchange=False)
        g.flush()                            # enqueue the held-back EDT of this frame's updateEsdf
        ev[i][1].record(stream)
        g.synchronize(); torch.cuda.synchronize(dev)
        lat_wall.append((time.perf_counter() - t) * 1e3)
    lat_gpu = [a.elapsed_time(b) for a, b in ev]
    pct = lambda v: {"mean": round(float(np.mean(v)), 4), "p50": round(float(np.percentile(v, 50)), 4), "p99": round(float(np.percentile(v, 99)), 4)}
    latency = {"frames": n2, "wall_ms": pct(lat_wall), "gpu_ms": pct(lat_gpu),
               "note": "one frame at a time with a synchronize after each (the ESDF distance transform then runs as its own launch)"}

    # per-kernel durations with hipEvent pairs on the mapper stream, same frames, mesh included
    g.set_profiling(True)
    counts_acc = {}
    for i in range(n2):
        step(base + i, mesh=True, exchange=False)
        if i % 10 == 0:
            c = g.counters()
            for k_, v_ in c.items():
                counts_acc.setdefault(k_, []).append(v_)
    prof = g.profile()
    g.set_profiling(False)
    counts = {k_: float(np.mean(v_)) for k_, v_ in counts_acc.items()}
    import re
    short = lambda k_: re.match(r"\s*(k_[a-z_0-9]+)", k_).group(1)
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_round.sh)
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}
    kern = {}
    ev_pair = prof.pop("_empty_event_pair", None)
    empty_pair_us = (ev_pair["total_ms"] / ev_pair["count"] * 1e3) if ev_pair else 0.0
    # A hipEvent pair around ONE launch adds its own cost to the span.  Calibration: the launches of the timed frame (everything
    # but the mesh and the stand-alone EDT, which rides in k_mark_view there) must add up to the un-instrumented frame time
    # measured above; the excess is the instrumentation, split evenly over those launches.
    frame_launches = [k_ for k_ in prof if not (short(k_).startswith("k_mesh") or short(k_).startswith("k_esdf_edt"))]
    raw_sum_us = sum(prof[k_]["total_ms"] / n2 * 1e3 for k_ in frame_launches)
    n_frame_launches = sum(prof[k_]["count"] / n2 for k_ in frame_launches)
    ev_overhead_us = max(0.0, (raw_sum_us - ms_per_step * 1e3) / max(1.0, n_frame_launches))
    for k_, v_ in prof.items():
        us = max(0.1, v_["total_ms"] / v_["count"] * 1e3 - ev_overhead_us)
        ab = algorithmic_bytes(short(k_), counts, rows, cols)
        kern[short(k_)] = {"avg_us": us, "launches_per_frame": v_["count"] / n2, "algorithmic_bytes": int(ab),
                           "achieved_GBps": (ab / (us * 1e-6) / 1e9) if us > 0 else 0.0,
                           "hbm_traffic_bytes": pmc.get(short(k_), {}).get("hbm_bytes_per_launch")}
    hot = [k_ for k_ in kern if not k_.startswith("k_mesh")]
    # No kernel dominates the frame by time (six launches of 7-11 us each, profiles/*_kernel_stats.csv), so the HBM roofline
    # is quoted for the kernel that moves the most bytes; the longest one is named beside it and every kernel is listed.
    dom = max(hot, key=lambda k_: kern[k_]["algorithmic_bytes"] * kern[k_]["launches_per_frame"])
    longest = max(hot, key=lambda k_: kern[k_]["avg_us"] * kern[k_]["launches_per_frame"])
    dom_bytes = kern[dom]["algorithmic_bytes"]
    dom_us = kern[dom]["avg_us"]
    achieved = kern[dom]["achieved_GBps"]
    traffic = kern[dom]["hbm_traffic_bytes"]
    frame_bytes = sum(kern[k_]["algorithmic_bytes"] * kern[k_]["launches_per_frame"] for k_ in hot)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_us": round(dom_us, 3), "event_pair_overhead_us": round(ev_overhead_us, 3), "empty_event_pair_us": round(empty_pair_us, 3),
                "frame": {"algorithmic_bytes": int(frame_bytes), "achieved": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                          "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                "longest_kernel": {"kernel": longest, "avg_launch_us": round(kern[longest]["avg_us"], 3),
                                   "achieved": round(kern[longest]["achieved_GBps"], 2), "frac": round(kern[longest]["achieved_GBps"] / HBM_PEAK_GBS, 5)},
                "note": "kernel = the non-mesh kernel with the most algorithmic HBM bytes per frame (no kernel dominates by time: four "
                        "dependent launches of 6-11 us each; k_integrate_color also carries the ESDF site marking, k_mark_view the "
                        "held-back EDT of the previous update -- DESIGN.md 2.4); durations = span of a hipEvent pair around each launch on the mapper stream minus `event_pair_overhead_us`, the "
                        "instrumentation cost per launch calibrated so that the frame's launches add up to the un-instrumented frame time; "
                        "compare rocprofv3's kernel-trace averages in profiles/*_kernel_stats.csv. 640x480 @ 0.05 m moves ~15 MB/frame, so every kernel is bound by its "
                        "dependent-access chain and launch cost rather than by HBM bytes (DESIGN.md 2); per-kernel achieved GB/s under "
                        "`kernels`, whole frame under `frame`",
                "components_note": "ms_components are measured per call in isolation: the EDT of updateEsdf is held back and runs "
                                   "inside the next depth frame's first launch, so `esdf` shows the marking launch only and "
                                   "the sum of the three is not ms_per_step"}

    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        # 8 OpenMP threads: the oracle's parallel regions are one frame's blocks (a few hundred tasks), which stop scaling
        # there (1 thread 7.6 ms, 8 threads 3.4 ms, 256 threads 620 ms per frame on the EPYC 9575F host -- BASELINE.md 3)
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        po = oracle.OrcParams()
        for name, _ in oracle.OrcParams._fields_:
            setattr(po, name, getattr(g.params, name))
        o = oracle.OracleMap(po)
        nf = max(2, args.cpu_frames)
        for k in range(2):    # warm the map like the GPU warm-up does
            d, c_, T = host_frames[k]
            o.integrate_depth(d, T, cam); o.integrate_color(c_, T, cam); o.update_esdf()
        t = time.perf_counter()
        k = 0
        while k < nf or (time.perf_counter() - t < args.cpu_seconds and k < 20000):     # a bounded sample: ~10-30 s of CPU work
            d, c_, T = host_frames[(2 + k) % nu]
            o.integrate_depth(d, T, cam); o.integrate_color(c_, T, cam); o.update_esdf()
            k += 1
        nf = k
        cdt = time.perf_counter() - t
        cpu = {"value": round(nf / cdt, 3), "unit": "frames/s", "cores": int(oracle.num_threads()), "kind": "port",
               "ms_per_frame": round(cdt / nf * 1e3, 2),
               "sample": "%d frames of the same 640x480 sequence, TSDF+Color+ESDF, oracle/nvblox_oracle.c, OpenMP with %d threads" % (nf, int(oracle.num_threads()))}

    out = {
        "metric": "frames/s, TSDF+Color+ESDF integrate per frame, synthetic Replica-like 640x480 @0.05m",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic Replica-like room (SURVEY 8d), 640x480 depth+colour, 0.05 m voxels, "
                               "fuser.yaml params, TSDF+Color+ESDF every frame (mesh timed separately)",
                   "cameras_per_gpu": 1, "parallelism": "one camera per GPU, RCCL all-gather of dirty block indices" if world > 1 else "single GPU",
                   "unique_frames": nu},
        "ms_per_frame": round(ms_per_step, 4),
        "ms_components": {k_: round(v_, 4) for k_, v_ in comp.items()},
        "frame_latency": latency,
        "readme_rtx5090_ms": README_RTX5090_MS,
        "speedup_vs_readme_rtx5090_tsdf_color_esdf": round(0.7 / ms_per_step, 2),
        "per_frame_counts": {k_: round(v_, 1) for k_, v_ in counts.items()},
        "kernels": {k_: {"avg_us": round(v_["avg_us"], 3), "launches_per_frame": round(v_["launches_per_frame"], 2),
                         "algorithmic_bytes": v_["algorithmic_bytes"], "achieved_GBps": round(v_["achieved_GBps"], 1),
                         "hbm_traffic_bytes": v_["hbm_traffic_bytes"]} for k_, v_ in kern.items()},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
