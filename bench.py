#!/usr/bin/env python3
"""bench.py -- throughput of the nvblox hot path on MI355X, one JSON line per run (rank 0).

Workloads (BASELINE.json `configs`; synthetic stand-ins of SURVEY.md 8d [D] -- no dataset on disk, no network):
  camera   configs[1], the metric's configuration (default): Replica-like room, 640x480 depth + colour, 0.05 m voxels, fuser.yaml
           parameters; step = integrateDepth + integrateColor + updateEsdf of one frame, inputs resident in HBM.  N > 1 GPUs: one
           camera per GPU (45 deg yaw offsets, configs[3]), all-gather of dirty block indices over RCCL before every ESDF sweep.
  multicam configs[3] on ONE GPU, --cameras C (1..8): the reference's own multi-camera mode (<= 4 cameras feed one mapper,
           nvblox_node.hpp:298-332); step = C depth frames + C colour frames + one ESDF update; value = camera-frames/s.
  decay    configs[2]: Redwood-like room with a moving box, MappingType::kDynamic: freespace layer + dynamics detection + mask
           clean-up + occupancy mapper for the dynamic part, invalid_depth_decay 0.8, TSDF / occupancy decay every 6th frame.
  lidar    configs[4]: 1024x64 spinning LiDAR, 0.10 m voxels, 200 m; step = one scan.  N > 1 GPUs: azimuth sectors.

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides (max over ranks);
blocks are repeated until >= 1 s has been timed (`repeats`), `ms_per_step` / `value` come from the MEDIAN block (camera / multicam: the
EXPLORING figure, mean over complete loops of the unique poses on a map emptied per loop -- see main_camera).
`roofline` is quoted for the LONGEST kernel of the step (by time); the kernel with the most bytes is listed beside it.
"""
import argparse
import gc
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

README_RTX5090_MS = {"tsdf": 0.1, "color": 0.3, "esdf": 0.3, "mesh": 0.3, "dynamics": 0.7}   # /root/reference README.md:69-106 (Replica)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MIN_TIMED_MS = float(os.environ.get("NVBX_BENCH_MIN_MS", "1000"))     # (env: profiling runs under rocprofv3 keep their traces small)
# blocks of exactly K steps are repeated until at least this much has been timed (the driver's busy sampler then sees the work)


def short(name):
    m = re.match(r"\s*(?:void\s+)?(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name.strip()


def algorithmic_bytes(kernel, c, rows, cols, sub_ray=4, sub_trace=4, n_cam=1, trace_in_mark_view=False, fused=False):
    """SURVEY.md 8(d) algorithmic bytes of ONE launch of `kernel`, from the measured per-launch counts `c`.
    trace_in_mark_view: colour deferral, the sphere tracing of the previous colour frame rides in the view-marking launch;
    fused: ... and so do that frame's candidate discovery and ESDF site marking, while its colour integration and distance transform share
    the TSDF update's launch (k_integrate_tsdf_color) -- two launches per frame, DESIGN.md 2.8."""
    B = 4096
    Nv, Nc, Na = c.get("tsdf_blocks_in_view", 0), c.get("color_blocks_updated", 0), c.get("blocks_allocated", 0)
    Nu, Ne = c.get("esdf_columns_marked", 0), c.get("esdf_blocks_swept", 0)
    color_only = n_cam * (rows * cols * 3 + (rows // sub_trace) * (cols // sub_trace) * 4) + Nc * B * 2      # colour image + synthetic depth read + colour RMW of the band blocks
    marking = Nu * 2 * B + Nu * 24                                                                            # TSDF z-band (k_z = 2 blocks) of the re-marked columns read + three mask words written
    edt = Ne * (512 * 2 + 121 * (16 + 8))                                                                     # plane RMW + 11x11 neighbour hash entries and site masks per swept block
    if kernel.startswith("k_integrate_tsdf_color_pair") or kernel.startswith("k_mark_view_pair"):
        # two mappers' launches in one grid (nvbx_integrate_depth_pair): the first mapper's bytes (counts `c`) + the second's (counts under "b_...": the
        # foreground occupancy mapper -- its half of the depth image, its few blocks, the marking pass / distance transform of its held-back update)
        base = "k_integrate_tsdf_color" if kernel.startswith("k_integrate") else "k_mark_view"
        cb = {k_[2:]: v_ for k_, v_ in c.items() if k_.startswith("b_")}
        cb["color_blocks_updated"] = 0
        return (algorithmic_bytes(base, c, rows, cols, sub_ray, sub_trace, n_cam, trace_in_mark_view, fused) +
                algorithmic_bytes(base, cb, rows, cols, sub_ray, sub_trace, n_cam, False, False) - (n_cam * (rows * cols * 3 + (rows // sub_trace) * (cols // sub_trace) * 4) if base == "k_integrate_tsdf_color" else 0))
    if kernel.startswith("k_integrate_tsdf_color"):
        # TSDF update of this frame + colour integration of the held-back frame (candidates arrive as 16-byte records) + the distance transform
        return algorithmic_bytes("k_integrate_tsdf", c, rows, cols, sub_ray, sub_trace, n_cam) + color_only + Nc * 16 + edt
    if kernel.startswith("k_integrate_tsdf"):
        # (LiDAR: the blocks the beam-centric launch has taken are skipped by this one -- it reads their records, not their voxels)
        Ns = c.get("lidar_blocks_beam_centric", 0)
        return n_cam * rows * cols * 4 + Nv * (12 + 8) + (Nv - Ns) * B * 2
    if kernel.startswith("k_lidar_sparse"):
        # per block in view: record + class byte; per block it updates: its footprint of the range image (<= 64 pixels + quads), the beams'
        # table entries and ~21 voxels read + written (8-byte voxels moved as 32-byte sectors)
        Ns = c.get("lidar_blocks_beam_centric", 0)
        return Nv * 17 + Ns * (64 * 4 + 130 * 4 + 21 * 64)
    if kernel.startswith("k_mark_view_grid"):
        # LiDAR view calculation over the dense grid: sub-sampled range image read; per block in view one byte of the grid and one
        # byte of the coarse cell map stored (visits beyond the first store the same bytes again: not counted)
        return (rows // sub_ray) * (cols // sub_ray) * 4 + Nv * 2
    if kernel.startswith("k_scan_view_grid"):
        # the coarse cell map read (and its touched bytes cleared), the set bytes read and cleared, one 16-byte record per block written
        return c.get("view_grid_cells", 0) + Nv * (1 + 1 + 16)
    if kernel.startswith("k_resolve_view"):
        return Nv * (16 + 16 + 4 + 4)                  # record read, hash entry read, entry stamp + the record's slot written
    if kernel.startswith("k_mark_view"):
        own = n_cam * (rows // sub_ray) * (cols // sub_ray) * 4 + Nv * 16 * 2           # sub-sampled depth read + one hash entry RMW per block in view
        if trace_in_mark_view or fused:
            own += algorithmic_bytes("k_sphere_trace", c, rows, cols, sub_ray, sub_trace, n_cam)
        if fused:
            own += Na * 16 + Nc * 16 + marking          # flags + Index3D of every allocated slot, one record per candidate; the marking pass
        return own
    if kernel.startswith("k_integrate_color"):
        # (the band vote is a per-block flag: no TSDF read) plus the ESDF site marking that rides in the same launch
        return color_only + marking
    if kernel.startswith("k_sphere_trace"):
        return n_cam * (rows // sub_trace) * (cols // sub_trace) * 4 + Nc * B           # synthetic depth write + TSDF blocks read once
    if kernel.startswith("k_esdf_mark"):
        return marking
    if kernel.startswith("k_esdf_edt"):
        return edt
    if kernel.startswith("k_mesh"):
        return int(c.get("mesh_blocks_updated", 0) * B * (1.42 + 1.0) + c.get("mesh_vertices", 0) * 28 + c.get("mesh_triangles", 0) * 12)
    if kernel.startswith("k_decay"):
        return Na * (B * 2 + 16)                                     # every projective voxel read + written, flags
    if kernel.startswith("k_update_freespace"):
        return Nv * (B + 2 * 512 * 16)                               # TSDF read, freespace voxels (16 B) read + written
    if kernel.startswith("k_detect_dynamics") or kernel.startswith("k_split_depth") or kernel.startswith("k_mask_zmin"):
        return rows * cols * (4 + 4 + 1)
    if kernel.startswith("k_stage_color"):
        return 0                 # OVERHEAD, not algorithm (SURVEY 8d has no such term): overhead_bytes() below; absent when the images live in library frames
    if kernel.startswith("k_dyn_detect_union"):
        return rows * cols * (4 + 1 + 4 + 4)          # depth read, mask written, label + nearest-depth images touched
    if kernel.startswith("k_dyn_filter_split"):
        return rows * cols * (4 + 1 + 4 + 4 + 8 + 12)     # depth, mask, label, nearest depth read; two depth images written; the next call's three arrays reset
    if kernel.startswith("k_cc_"):
        return rows * cols * 8
    if kernel.startswith("k_save_stamps") or kernel.startswith("k_reinsert"):
        return Na * 32
    return 0


def overhead_bytes(kernel, rows, cols, n_cam=1):
    """Bytes a launch moves that SURVEY 8d's formulas do not contain: the staged form's copy of raw-pointer colour images (read + written)."""
    return n_cam * rows * cols * 3 * 2 if kernel.startswith("k_stage_color") else 0


def pmc_source(workload=None):
    """Where `traffic` comes from: the committed table of the last PMC passes (never measured inside this run: counters need rocprofv3)."""
    try:
        meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("_meta") or {}
    except Exception:
        meta = {}
    w = (meta.get("by_workload") or {}).get(workload) or meta
    return "profiles/pmc_latest.json[%s]@%s (%s)" % (workload or "camera", w.get("commit_when_summarised", w.get("commit", "unknown")), w.get("tag", "?"))


def load_pmc(workload):
    """HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_round.sh, tools/summarize_profile.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return {}
    if workload in pmc and isinstance(pmc[workload], dict) and not any(k.startswith("FETCH") for k in pmc[workload]):
        out = dict(pmc[workload])
        if workload == "camera" and "k_mesh" in pmc.get("camera_mesh", {}):       # (k_mesh: from the pass that updates the mesh every frame, --with-mesh)
            out["k_mesh"] = pmc["camera_mesh"]["k_mesh"]
        return out
    return pmc if workload == "camera" else {}


class Timer:
    """Blocks of exactly `steps` steps; each block bracketed by barrier(); max over ranks; repeated until MIN_TIMED_MS."""

    def __init__(self, torch, dist, dev, world):
        self.torch, self.dist, self.dev, self.world = torch, dist, dev, world

    def block(self, run, barrier, steps, first, before_block=None):
        if before_block is not None:
            barrier(); before_block()        # (outside the timed region: e.g. a fresh map for an exploring block)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            run(first + i)
        barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def run(self, run, barrier, steps, warmup, min_ms=MIN_TIMED_MS, max_repeats=20000, before_block=None, first=None):
        for i in range(warmup):
            run(i)
        dts = []
        first = warmup if first is None else first
        # The host loop is Python: a generation-2 pass of its cyclic garbage collector over the ~10^6 objects torch + numpy keep alive
        # takes ~40 ms and lands in whichever block crosses the allocation threshold (round 2's decay line: ONE block of three 3.8 x
        # slower in every run; traced to the host with --step-trace: no kernel of that step took longer than 32 us).  A C++ host has
        # no such pause, so the collector is parked for the timed region (NVBX_BENCH_KEEP_GC=1 keeps it on, to reproduce the outlier).
        park = os.environ.get("NVBX_BENCH_KEEP_GC") != "1"
        if park:
            gc.collect(); gc.freeze(); gc.disable()
        try:
            while True:
                dts.append(self.block(run, barrier, steps, first, before_block)); first += steps
                if (sum(dts) * 1e3 >= min_ms and len(dts) >= 3) or len(dts) >= max_repeats:     # (same decision on every rank: dt is all-reduced)
                    break
        finally:
            if park:
                gc.enable(); gc.unfreeze()
        return float(np.median(dts)), dts, first


def complete_loops(tags, dts, nu, steps):
    """Exploring bookkeeping: tags[i] = position in the loop over the nu unique poses at which timed block i started (0 = the map had just been
    emptied), dts[i] = its duration.  -> (blocks per loop, indices of the blocks that started a loop, indices of the first blocks of COMPLETE
    loops, durations of all blocks of complete loops -- or of every block if no loop was completed)."""
    per_loop = max(1, nu // steps)
    starts = [i for i, t in enumerate(tags) if t == 0]
    whole = [i for i in starts if i + per_loop <= len(dts) and all(tags[i + j] == j * steps for j in range(per_loop))]
    kept = [dts[i + j] for i in whole for j in range(per_loop)] or list(dts)
    return per_loop, starts, whole, kept


def block_stats(dts, steps):
    """per-step milliseconds of the timed blocks: median (the reported figure), mean, p99, min / max, how many blocks, total timed"""
    v = np.asarray(dts) / steps * 1e3
    return {"blocks": int(len(v)), "median": round(float(np.median(v)), 4), "mean": round(float(v.mean()), 4), "p99": round(float(np.percentile(v, 99)), 4),
            "min": round(float(v.min()), 4), "max": round(float(v.max()), 4), "timed_ms_total": round(float(np.sum(dts)) * 1e3, 1)}


def kernel_table(prof, counts, ms_per_step, n_steps, bytes_fn, pmc, exclude_from_calibration=("k_mesh",)):
    """hipEvent spans per launch -> {kernel: avg_us, launches_per_step, algorithmic bytes, achieved GB/s, PMC traffic}.
    A hipEvent pair around ONE launch adds its own cost to the span; calibration: the launches of the timed step must add up to the
    un-instrumented step time, the excess is split evenly over them."""
    ev_pair = prof.pop("_empty_event_pair", None)
    empty_pair_us = (ev_pair["total_ms"] / ev_pair["count"] * 1e3) if ev_pair else 0.0
    in_step = [k for k in prof if not short(k).startswith(exclude_from_calibration)]
    raw_sum_us = sum(prof[k]["total_ms"] / n_steps * 1e3 for k in in_step)
    n_launch = sum(prof[k]["count"] / n_steps for k in in_step)
    ev_overhead_us = max(0.0, (raw_sum_us - ms_per_step * 1e3) / max(1.0, n_launch))
    kern = {}
    for k, v in prof.items():
        us = max(0.1, v["total_ms"] / v["count"] * 1e3 - ev_overhead_us)
        ab = bytes_fn(short(k), counts)
        name = short(k)
        if name in kern:       # template instances of one kernel (e.g. two depth sources): merge
            o = kern[name]; n0, n1 = o["launches_per_step"], v["count"] / n_steps
            o["avg_us"] = (o["avg_us"] * n0 + us * n1) / (n0 + n1); o["launches_per_step"] = n0 + n1
            continue
        kern[name] = {"avg_us": us, "launches_per_step": v["count"] / n_steps, "algorithmic_bytes": int(ab),
                      "achieved_GBps": (ab / (us * 1e-6) / 1e9) if us > 0 else 0.0,
                      "hbm_traffic_bytes": pmc.get(name, {}).get("hbm_bytes_per_launch")}
    return kern, ev_overhead_us, empty_pair_us


def roofline_of(kern, ms_per_step, ev_overhead_us, empty_pair_us, note, skip=("k_mesh",)):
    hot = [k for k in kern if not k.startswith(skip) and kern[k]["algorithmic_bytes"] > 0]
    # the longest kernel of the step by time.  The camera frame has three launches within half a microsecond of each other (view marking,
    # sphere tracing, colour), less than the event timing resolves run to run: kernels within 7 % of the longest count as tied and the tie
    # goes to the one that moves the most algorithmic bytes (all of them are listed under `longest_tied`, every kernel under `kernels`)
    t_of = lambda k: kern[k]["avg_us"] * kern[k]["launches_per_step"]
    t_max = max(t_of(k) for k in hot)
    tied = sorted((k for k in hot if t_of(k) >= 0.93 * t_max), key=t_of, reverse=True)
    longest = max(tied, key=lambda k: kern[k]["algorithmic_bytes"] * kern[k]["launches_per_step"])
    most = max(hot, key=lambda k: kern[k]["algorithmic_bytes"] * kern[k]["launches_per_step"])
    frame_bytes = sum(kern[k]["algorithmic_bytes"] * kern[k]["launches_per_step"] for k in hot)
    k = kern[longest]
    return {"bound": "hbm", "kernel": longest, "achieved": round(k["achieved_GBps"], 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(k["achieved_GBps"] / HBM_PEAK_GBS, 5), "traffic": k["hbm_traffic_bytes"],
            # the same fraction with the MEASURED HBM bytes (PMC FETCH_SIZE + WRITE_SIZE per launch) in place of the SURVEY 8d formula's:
            # what the kernel really moves per second against the peak (the formula credits e.g. a write of every voxel of a block in view)
            "frac_by_traffic": (round(k["hbm_traffic_bytes"] / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if k["hbm_traffic_bytes"] else None),
            "algorithmic_bytes_per_launch": int(k["algorithmic_bytes"]), "avg_launch_us": round(k["avg_us"], 3),
            "launches_per_step": round(k["launches_per_step"], 2),
            "longest_tied": {t: round(kern[t]["avg_us"], 3) for t in tied},
            "event_pair_overhead_us": round(ev_overhead_us, 3), "empty_event_pair_us": round(empty_pair_us, 3),
            "step": {"algorithmic_bytes": int(frame_bytes), "achieved": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                     "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "most_bytes_kernel": {"kernel": most, "avg_launch_us": round(kern[most]["avg_us"], 3), "achieved": round(kern[most]["achieved_GBps"], 2),
                                  "frac": round(kern[most]["achieved_GBps"] / HBM_PEAK_GBS, 5), "traffic": kern[most]["hbm_traffic_bytes"]},
            # context for `peak` (the paper figure): what a bare stream over 8-byte voxels reaches on this GPU, measured once
            "measured_stream_ceiling": {"read_modify_write_GBps": 5800, "read_GBps": 6450, "source": "tools/stream_ceiling.hip (profiles/r02z_stream_ceiling.txt)"},
            # the 640x480 working set (10-40 MB) sits in the 256 MiB last-level cache / the 32 MiB of L2, so the same bytes against the L2 fabric's
            # rate too (MI355X_MICROARCH.md: 34.5 TB/s) -- both fractions are small because the launches are dependent-access chains, not streams
            "l2_fabric": {"peak_GBps": 34500, "frac": round(k["achieved_GBps"] / 34500.0, 5), "step_frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / 34500.0, 5)},
            "counters": "profiles/r06_request_counters.json (L2 requests, L1 accesses, VALU / LDS / VMEM wave-instructions per launch); what FETCH_SIZE / WRITE_SIZE mean for these access patterns: profiles/r06_pmc_calibration.json",
            "note": note}


def kernels_json(kern):
    return {k: {"avg_us": round(v["avg_us"], 3), "launches_per_step": round(v["launches_per_step"], 2), "algorithmic_bytes": v["algorithmic_bytes"],
                "achieved_GBps": round(v["achieved_GBps"], 1), "hbm_traffic_bytes": v["hbm_traffic_bytes"]} for k, v in kern.items()}


def copy_params(oracle, p):
    po = oracle.OrcParams()
    for name, _ in oracle.OrcParams._fields_:
        setattr(po, name, getattr(p, name))
    return po


def cpu_sample(step_cpu, seconds, min_steps, unit, what, oracle):
    """The oracle (a port: there is no CPU path in the reference) on a bounded sample of the same workload."""
    t = time.perf_counter(); k = 0
    while k < min_steps or (time.perf_counter() - t < seconds and k < 100000):
        step_cpu(k); k += 1
    dt = time.perf_counter() - t
    return {"value": round(k / dt, 3), "unit": unit, "cores": int(oracle.num_threads()), "kind": "port", "ms_per_step": round(dt / k * 1e3, 3),
            "sample": "%d %s, oracle/nvblox_oracle.c, OpenMP with %d threads" % (k, what, int(oracle.num_threads()))}


def map_parity(M, g, o, oracle):
    """The HIP map `g` against the checker map `o` (oracle.OracleMap) fed the same calls: block index sets (bit-exact), TSDF distance / weight
    (north_star: 1e-4), colour (+-1 LSB) and colour weight, every ESDF voxel field (exact), the 2-D slice image (1e-4 m).  Test
    infrastructure: called outside every timed region (bench.py's `parity` block, tests/test_gpu_round4.py)."""
    out = {}
    it, io = g.block_indices(M.LAYER_TSDF), o.block_indices(oracle.L_TSDF)
    out["blocks"] = int(len(it)); out["blocks_checker"] = int(len(io))
    out["index_sets_equal"] = bool(np.array_equal(it, io))
    ok = out["index_sets_equal"]
    if ok and len(it):
        bg, _ = g.get_blocks(M.LAYER_TSDF, it)
        bo = np.stack([o.get_block(oracle.L_TSDF, i) for i in io])
        out["max_abs_tsdf"] = float(max(np.abs(bg["distance"].astype(np.float64) - bo["distance"]).max(), np.abs(bg["weight"].astype(np.float64) - bo["weight"]).max()))
    ic, ico = g.block_indices(M.LAYER_COLOR), o.block_indices(oracle.L_COLOR)
    out["color_blocks"] = int(len(ic)); out["color_index_sets_equal"] = bool(np.array_equal(ic, ico))
    if out["color_index_sets_equal"] and len(ic):
        bg, _ = g.get_blocks(M.LAYER_COLOR, ic)
        bo = np.stack([o.get_block(oracle.L_COLOR, i) for i in ico])
        out["max_color_lsb"] = int(max(np.abs(bg[f].astype(np.int32) - bo[f].astype(np.int32)).max() for f in ("r", "g", "b")))
        out["max_abs_color_weight"] = float(np.abs(bg["weight"].astype(np.float64) - bo["weight"]).max())
    ie, ieo = g.block_indices(M.LAYER_ESDF), o.block_indices(oracle.L_ESDF)
    out["esdf_blocks"] = int(len(ie)); out["esdf_index_sets_equal"] = bool(np.array_equal(ie, ieo))
    if out["esdf_index_sets_equal"] and len(ie):
        bg, _ = g.get_blocks(M.LAYER_ESDF, ie)
        bo = np.stack([o.get_block(oracle.L_ESDF, i) for i in ieo])
        out["esdf_voxels_differing"] = int(sum(int((bg[f] != bo[f]).sum()) for f in ("squared_distance_vox", "parent_direction", "is_inside", "observed", "is_site")))
    sg, ag = g.esdf_slice_image(); so, ao = o.esdf_slice_image()
    out["slice_shape"] = [int(sg.shape[0]), int(sg.shape[1])]
    out["slice_shapes_equal"] = bool(sg.shape == so.shape and np.array_equal(ag, ao))
    if out["slice_shapes_equal"] and sg.size:
        out["max_abs_slice_m"] = float(np.abs(sg.astype(np.float64) - so).max())
    out["ok"] = bool(ok and out["color_index_sets_equal"] and out["esdf_index_sets_equal"] and out["slice_shapes_equal"] and
                     out.get("max_abs_tsdf", 0.0) <= 1e-4 and out.get("max_color_lsb", 0) <= 1 and out.get("max_abs_color_weight", 0.0) <= 1e-4 and
                     out.get("esdf_voxels_differing", 0) == 0 and out.get("max_abs_slice_m", 0.0) <= 1e-4)
    return out


def exploring_parity(M, g, gpu_step, drain, n_frames, checker_step, steps, oracle, threads=8):
    """The sequence the headline times, run once more OUTSIDE the timed region and compared with the checker at its end: clear(), every pose
    of the loop once (depth, colour, updateEsdf per frame through `gpu_step(k)`), a drain (`drain()`: what ends a timed block -- synchronize,
    which replays whatever the mapper holds back) after every `steps` frames; then the CPU checker is fed the same frames in plain call order."""
    g.clear()
    for k in range(n_frames):
        gpu_step(k)
        if (k + 1) % steps == 0:
            drain()
    drain()
    oracle.set_num_threads(min(threads, os.cpu_count() or 1))
    o = oracle.OracleMap(copy_params(oracle, g.params))
    t = time.perf_counter()
    for k in range(n_frames):
        checker_step(o, k)
    out = map_parity(M, g, o, oracle)
    out["steps_compared"] = int(n_frames); out["drain_every"] = int(steps); out["checker_s"] = round(time.perf_counter() - t, 2)
    out["what"] = ("one more exploring loop exactly as timed (map emptied, %d steps of depth + colour + updateEsdf, a drain every %d steps) against "
                   "oracle/nvblox_oracle.c fed the same frames in plain call order; outside the timed region" % (n_frames, steps))
    return out


def launch_command(n, argv, port=None):
    """`python bench.py --gpus N ...` without WORLD_SIZE in the environment re-executes itself as N ranks, one per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ..."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launch_command(n, argv)
    if os.environ.get("NVBX_BENCH_LAUNCH_DRY") == "1":     # (tests: show the command, start nothing)
        print(" ".join(cmd)); return 0
    return subprocess.call(cmd, env=env)


def init_dist(args):
    import torch
    import torch.distributed as dist
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
    return torch, dist, rank, world, local_rank, dev


def finish_dist(dist, world):
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


# ====================================================================================================== lidar (configs[4])
def main_lidar(args):
    """One step = integrateDepth of one 1024x64 LiDAR scan (range image resident in HBM), 0.10 m voxels, 200 m.  N > 1 GPUs: the scan's
    azimuth range is cut into N sectors, rank r integrates the beams of sector r into its own map (strong scaling of one sensor; the
    sector maps are disjoint wedges apart from the blocks on the cuts -- DESIGN.md 6)."""
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    torch, dist, rank, world, local_rank, dev = init_dist(args)
    lidar = S.SPINNING_LIDAR
    sc = S.LidarScene()
    nu = max(2, min(args.unique_frames, 16))
    host = []
    for i in range(nu):
        T = S.lidar_pose(i, 400)
        img = S.render_lidar(sc, T, lidar, max_range=200.0)
        if world > 1:                                  # this rank's azimuth sector: the other beams are "no return"
            cols = img.shape[1]; lo, hi = rank * cols // world, (rank + 1) * cols // world
            mask = np.zeros(cols, bool); mask[lo:hi] = True
            img = np.where(mask[None, :], img, np.float32(0.0)).astype(np.float32)
        host.append((img, T))
    scans = [(torch.from_numpy(r).to(dev), T) for r, T in host]
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    p = M.default_params(voxel_size=0.1, lidar_max_integration_distance_m=200.0, raycast_subsampling_factor=2)
    g = M.Mapper(p, device=local_rank, block_capacity=1 << 19, stream=stream.cuda_stream)
    largs = [g.prepare_lidar(r, T, lidar) for r, T in scans]

    def barrier():
        g.synchronize(); torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    tm = Timer(torch, dist, dev, world)
    dt, dts, nxt = tm.run(lambda i: g.integrate_prepared(largs[i % nu]), barrier, args.steps, args.warmup)
    ms = dt / args.steps * 1e3
    # EXPLORING beside the revisit figure above: every timed block starts from an EMPTY map (clear() outside the timed region) and
    # integrates the nu scans of the pose loop once -- the first scan allocates every block in view (~112 k hash inserts and slot pops), the
    # following ones the blocks the moving sensor newly sees
    if args.profile_run:         # (under rocprofv3: nothing but the timed scan is launched, the trace's averages are the revisit scan's)
        dt_x, dts_x = dt / args.steps * nu, [dt / args.steps * nu]
    else:
        dt_x, dts_x, _ = tm.run(lambda i: g.integrate_prepared(largs[i % nu]), barrier, nu, 0, min_ms=MIN_TIMED_MS / 2, before_block=g.clear, first=0)
        for i in range(nu):      # (the map is complete again for the profiling passes below)
            g.integrate_prepared(largs[i % nu])
    ms_exploring = dt_x / nu * 1e3
    if rank != 0:
        return finish_dist(dist, world)
    n2 = min(args.steps, 40)
    g.set_profiling(True)
    nv = []; nsp = []
    for i in range(n2):
        g.integrate_prepared(largs[i % nu]); cc_ = g.counters(); nv.append(cc_["tsdf_blocks_in_view"]); nsp.append(cc_["lidar_blocks_beam_centric"])
    prof = g.profile(); g.set_profiling(False)
    c = g.counters(); counts = dict(c); counts["tsdf_blocks_in_view"] = float(np.mean(nv)); counts["lidar_blocks_beam_centric"] = float(np.mean(nsp))
    rows, cols = lidar[1], lidar[0]
    # cells of the view grid's coarse map (tsdf.hip launch_view_grid: box = range / block size + 2 blocks each way, z from the elevation range of a level sensor)
    reach = 200.0 / (8 * p.voxel_size); Hh = int(np.ceil(reach)) + 2; Hz = min(Hh, int(np.ceil(reach * np.sin(max(abs(lidar[3]), abs(lidar[4]))))) + 2)
    counts["view_grid_cells"] = ((2 * Hh + 1 + 3) // 4) ** 2 * ((2 * Hz + 1 + 3) // 4)
    kern, evo, emp = kernel_table(prof, counts, ms, n2, lambda k, cc: algorithmic_bytes(k, cc, rows, cols, sub_ray=2), load_pmc("lidar"))
    # the timed loop once more against the checker (outside the timed region): the nu scans from an empty map; view sets per scan, block index
    # set at the end, and every voxel of a seeded sample of the blocks
    parity = None
    if not args.no_parity and world == 1:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        o = oracle.OracleMap(copy_params(oracle, g.params))
        g.clear(); views_equal = True; t_par = time.perf_counter()
        for i in range(nu):
            g.integrate_prepared(largs[i]); o.integrate_lidar_depth(host[i][0], host[i][1], lidar)
            vg_ = np.asarray(g.last_view()).reshape(-1, 3); vo_ = np.asarray(o.last_view()).reshape(-1, 3)
            views_equal = views_equal and len(vg_) == len(vo_) and set(map(tuple, vg_.tolist())) == set(map(tuple, vo_.tolist()))
        ig, io = g.block_indices(M.LAYER_TSDF), o.block_indices(oracle.L_TSDF)
        same = bool(np.array_equal(ig, io)); worst = 0.0; n_cmp = 0
        if same and len(io):
            sel = io[np.random.default_rng(5).choice(len(io), size=min(4096, len(io)), replace=False)]
            bg, found = g.get_blocks(M.LAYER_TSDF, sel)
            for k_, idx in enumerate(sel):
                b = o.get_block(oracle.L_TSDF, idx)
                worst = max(worst, float(np.abs(bg[k_]["distance"].astype(np.float64) - b["distance"]).max()), float(np.abs(bg[k_]["weight"].astype(np.float64) - b["weight"]).max()))
            n_cmp = len(sel); same = same and bool(found.all())
        parity = {"scans": nu, "views_equal_every_scan": bool(views_equal), "blocks": int(len(ig)), "blocks_checker": int(len(io)), "index_sets_equal": same,
                  "blocks_compared_voxelwise": int(n_cmp), "max_abs_tsdf": worst, "ok": bool(views_equal and same and worst <= 1e-4), "checker_s": round(time.perf_counter() - t_par, 2),
                  "what": "the %d-scan loop from an empty map once more, every scan mirrored on oracle/nvblox_oracle.c: blocks in view per scan (sets), the "
                          "block index set at the end, every voxel of a seeded sample of blocks; outside the timed region (all blocks of two scans: "
                          "tests/test_gpu_full_size.py)" % nu}
        for i in range(nu):      # (the map as the profiling passes left it)
            g.integrate_prepared(largs[i % nu])
    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        o = oracle.OracleMap(copy_params(oracle, g.params))
        o.integrate_lidar_depth(host[0][0], host[0][1], lidar)          # warm the map like the GPU warm-up does
        cpu = cpu_sample(lambda k: o.integrate_lidar_depth(host[(1 + k) % nu][0], host[(1 + k) % nu][1], lidar), args.cpu_seconds, 2, "scans/s",
                         "scans of the same 1024x64 sequence (0.10 m voxels, 200 m)", oracle)
    out = {"metric": "scans/s, LiDAR projective TSDF integrate, synthetic 1024x64 @0.10m, 200 m", "value": round(args.steps / dt, 2),
           "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": len(dts), "ms_per_step": round(ms, 4),
           "ms_per_step_exploring": (None if args.profile_run else round(ms_exploring, 4)),
           "timing": {"value_is": "revisit: the %d poses of the loop cycled on the fully allocated map (146 k blocks)" % nu,
                      "exploring": "blocks of %d scans, each from an EMPTY map (allocation inside the timed region): %s" % (nu, json.dumps(block_stats(dts_x, nu)))},
           "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[4]: synthetic 1024x64 spinning LiDAR (SURVEY 8d), 0.10 m voxels, 200 m range, ray subsampling 2",
                      "parallelism": "azimuth sectors, one per GPU" if world > 1 else "single GPU"},
           "per_step_counts": {"tsdf_blocks_in_view": round(counts["tsdf_blocks_in_view"], 1), "lidar_blocks_beam_centric": round(counts["lidar_blocks_beam_centric"], 1),
                               "blocks_allocated": c["blocks_allocated"], "capacity_overflow": c["capacity_overflow"]},
           # the TSDF update as a whole (beam-centric far-field launch + dense launch): what the one-lane-per-voxel formulation of SURVEY 8d would
           # move for these blocks (every voxel of every block in view read + written) over the time the two launches take -- an EQUIVALENT rate:
           # the beam-centric launch touches ~20 voxels of a block, not 512, which is the point of it
           "tsdf_update": (lambda t_us, by: {"launches": [k_ for k_ in ("k_lidar_sparse", "k_integrate_tsdf") if k_ in kern], "us": round(t_us, 1),
                                             "dense_formula_bytes": int(by), "equivalent_GBps": round(by / (t_us * 1e-6) / 1e9, 1),
                                             "equivalent_frac": round(by / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})(
               sum(kern[k_]["avg_us"] for k_ in ("k_lidar_sparse", "k_integrate_tsdf") if k_ in kern),
               rows * cols * 4 + counts["tsdf_blocks_in_view"] * (20 + 4096 * 2)),
           "block_ms": [round(d / args.steps * 1e3, 4) for d in dts[:16]], "block_stats_ms_per_step": block_stats(dts, args.steps),
           "kernels": kernels_json(kern),
           "roofline": roofline_of(kern, ms, evo, emp, "longest kernel of the scan; durations = hipEvent spans on the mapper stream minus the calibrated "
                                   "instrumentation cost; compare profiles/*_lidar_kernel_stats.csv", skip=()),
           "cpu_baseline": cpu, "parity": parity}
    out["roofline"]["traffic_source"] = pmc_source("lidar")
    # What bounds the scan is VALU ISSUE, not HBM: wavefront instructions per launch from the committed SQ-counter pass (profiles/r05_lidar_sq_pmc.json,
    # tools/gpu_pmc.sh) over this run's launch durations, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614 G/s
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "r05_lidar_sq_pmc.json")))
        vi = {k_: {"valu_wave_instructions": int(sq[k_]["SQ_INSTS_VALU"]), "issue_frac": round(sq[k_]["SQ_INSTS_VALU"] / (v_["avg_us"] * 1e-6) / 614.4e9, 3)}
              for k_, v_ in kern.items() if k_ in sq and "SQ_INSTS_VALU" in sq[k_]}
        tot = sum(v_["valu_wave_instructions"] for v_ in vi.values())
        out["valu_issue"] = {"peak_G_per_s": 614.4, "kernels": vi, "scan_wave_instructions": int(tot), "scan_issue_frac": round(tot / (ms * 1e-3) / 614.4e9, 3),
                             "source": "profiles/r05_lidar_sq_pmc.json (SQ_INSTS_VALU per launch) over this run's durations",
                             "note": "the two TSDF-update launches issue vector instructions at 0.69-0.77 of the device's peak rate: the scan is VALU-bound, its HBM fraction is a consequence"}
    except Exception:
        pass
    # the scan as a whole by what its launches move (their own algorithmic bytes, not SURVEY 8d's one-lane-per-voxel formula)
    sb = sum(v_["algorithmic_bytes"] * v_["launches_per_step"] for v_ in kern.values())
    out["roofline"]["step"] = {"algorithmic_bytes": int(sb), "achieved": round(sb / (ms * 1e-3) / 1e9, 2), "frac": round(sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    print(json.dumps(out))
    finish_dist(dist, world)


# ====================================================================================================== decay (configs[2])
def main_decay(args):
    """Redwood-like dynamic scene, MappingType::kDynamic: per frame  detect dynamics -> remove small components -> split depth ->
    integrateDepth(static, TSDF + freespace, invalid-depth decay) -> integrateDepth(dynamic, occupancy) -> integrateColor -> updateEsdf
    (both mappers); every 6th frame decayTsdfExcludeLastView + decayOccupancyAllVoxels (5 Hz at 30 Hz input, nvblox_base.yaml:13-23)."""
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    torch, dist, rank, world, local_rank, dev = init_dist(args)
    assert world == 1, "the decay workload line is single-GPU"
    cam = S.REPLICA_LIKE_CAM; rows, cols = cam[5], cam[4]
    nu = max(6, min(args.unique_frames, 48))
    host = []
    for i in range(nu):
        sc = S.redwood_like_scene(i * 6)
        T = S.trajectory_pose(i * (200 // nu), 200, radius=1.2, height=1.4)
        d, rgb = S.render(sc, T, cam, max_range=5.0)
        host.append((d, rgb, T))
    depth_dev = [torch.from_numpy(d).to(dev) for d, _, _ in host]
    rgb_dev = [torch.from_numpy(c).to(dev) for _, c, _ in host]
    poses = [T for _, _, T in host]
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    # colour images in library-owned frames (the device memory of an nvblox::Image<Color> in the facade, csrc/frames.hip): a held-back frame is retained,
    # not copied; --staged-deferral: raw device pointers, one k_stage_color copy per held-back frame (round 4's line)
    use_frames = not args.staged_deferral and not args.no_color_deferral
    if use_frames:
        rgb_dev = [M.ColorFrame(rows, cols, 3, local_rank).write(t_, stream.cuda_stream) for t_ in rgb_dev]
    fs = dict(projective_layer_type=2, max_integration_distance_m=5.0, invalid_depth_decay_factor=0.8, tsdf_decay_factor=0.95,
              min_duration_since_occupied_for_freespace_ms=250)                     # nvblox_dynamics.yaml:11-18
    occ = dict(projective_layer_type=1, max_integration_distance_m=5.0)
    gs = M.Mapper(M.default_params(**fs), device=local_rank, block_capacity=1 << 15, stream=stream.cuda_stream)
    # both mappers on ONE stream, as nvblox::MultiMapper hands them out.  --own-stream (A/B, measured and not kept -- EXPERIMENTS.md): the dynamic
    # (occupancy) mapper on a stream of its own, the split image ordered with nvbx_mapper_wait_for: its launches are independent of the static
    # mapper's once the depth image is split, but an event record + wait per frame on the static mapper's stream costs more than the overlap saves
    stream_d = torch.cuda.Stream(dev) if args.own_stream else stream
    gd = M.Mapper(M.default_params(**occ), device=local_rank, block_capacity=1 << 13, stream=stream_d.cuda_stream)
    # cross-frame pipelining on the static mapper: the frame's first call, detect_dynamics, leaves held-back work alone, so the
    # colour frame / ESDF update of frame i are carried out by integrateDepth(i + 1) in two launches
    gs.set_color_deferral(not args.no_color_deferral, staged=True)           # (staged: the default form of a new mapper)
    gd.set_color_deferral(not args.no_color_deferral, staged=True)       # (an occupancy mapper has no colour: its updateEsdf alone is held back -- marking pass and distance transform ride in its next depth launches)
    pair = not args.no_depth_pair and not args.own_stream
    eye = np.eye(4, dtype=np.float32)
    mask = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    un = torch.empty((rows, cols), dtype=torch.float32, device=dev); ma2 = [torch.empty_like(un), torch.empty_like(un)]; ma = ma2[0]
    t_ms = [0]

    def step(i):
        k = i % nu
        ma = ma2[i & 1]                  # (two buffers: the split of frame i + 1 must not overwrite what the dynamic mapper reads in frame i)
        if args.separate_front_end:      # (A/B: the front end as three entry points = six launches + a memset)
            gs.detect_dynamics_into(depth_dev[k], poses[k], cam, 5.0, mask)
            gs.remove_small_components_inplace(mask, 2000)             # multi_mapper connected_mask_component_size_threshold (mapper_initialization.cpp:130)
            gs.split_depth_by_mask_into(depth_dev[k], mask, eye, cam, cam, 0.25, un, ma)
        else:                            # detect dynamics -> remove small components -> split: one call, three launches (what MultiMapper::integrateDepth runs)
            gs.dynamic_depth_split_into(depth_dev[k], poses[k], cam, 5.0, 2000, 0.25, mask, un, ma)
        gs.set_time_ms(t_ms[0]); t_ms[0] += 33
        gs.wait_for(gd)                  # (the dynamic mapper's frame i - 1 is over before the static mapper's stream goes on: frame i + 1's split rewrites that buffer)
        gd.wait_for(gs)                  # (`ma` is written)
        if pair:                         # both mappers' depth frames in two launches (nvbx_integrate_depth_pair: what MultiMapper::integrateDepth calls)
            gs.integrate_depth_pair(un, gd, ma, poses[k], cam)
        else:
            gs.integrate_depth(un, poses[k], cam)
            gd.integrate_depth(ma, poses[k], cam)
        gs.integrate_color(rgb_dev[k], poses[k], cam)
        gs.update_esdf(); gd.update_esdf()
        if i % 6 == 5:
            gs.decay_tsdf(True); gd.decay_occupancy()

    def barrier():
        gs.synchronize(); gd.synchronize(); torch.cuda.synchronize(dev)

    if args.step_trace:
        # diagnosis of slow blocks: every step waited for, wall time per step beside the pools' capacities (tools: VERDICT r02 weak #3)
        for i in range(args.warmup):
            step(i)
        barrier(); rec = []
        if os.environ.get("NVBX_BENCH_KEEP_GC") != "1":
            gc.collect(); gc.freeze(); gc.disable()
        calls = {}
        def wrap(obj, name):                 # every mapper call of the step: waited for and timed on its own
            fn = getattr(obj, name)
            def timed_call(*a, **kw):
                t = time.perf_counter(); r = fn(*a, **kw); barrier()
                calls.setdefault(cur[0], []).append((("gs." if obj is gs else "gd.") + name, round((time.perf_counter() - t) * 1e3, 3)))
                return r
            setattr(obj, name, timed_call)
        cur = [0]
        for o_ in (gs, gd):
            for nm in ("detect_dynamics_into", "remove_small_components_inplace", "split_depth_by_mask_into", "dynamic_depth_split_into", "integrate_depth", "integrate_color",
                       "update_esdf", "decay_tsdf", "decay_occupancy"):
                wrap(o_, nm)
        kern_of = {}
        for i in range(args.warmup, args.warmup + args.step_trace):
            cur[0] = i
            gs.set_profiling(True)           # (clears the spans: the profile below is this step's launches only)
            c0 = (gs.capacity, gd.capacity); t = time.perf_counter(); step(i); barrier()
            dt_ms = (time.perf_counter() - t) * 1e3
            rec.append((i, dt_ms, c0, (gs.capacity, gd.capacity)))
            if dt_ms > 1.0:
                pr = gs.profile(); pr.pop("_empty_event_pair", None)
                kern_of[i] = {"kernels_ms": {short(k_): round(v_["total_ms"], 3) for k_, v_ in pr.items()},
                              "esdf_window_voxels": gs.counters()["esdf_window_voxels"], "esdf_blocks_swept": gs.counters()["esdf_blocks_swept"]}
        gs.set_profiling(False)
        print(json.dumps({"slow_steps_kernels": kern_of}))
        w = np.array([r[1] for r in rec])
        worst = max(rec, key=lambda r: r[1])[0]
        print(json.dumps({"slowest_step_calls": {"step": worst, "calls_ms": calls.get(worst)}}))
        print(json.dumps({"step_trace": {"steps": len(rec), "median_ms": round(float(np.median(w)), 4), "mean_ms": round(float(w.mean()), 4),
                                         "p99_ms": round(float(np.percentile(w, 99)), 4),
                                         "slowest": [{"step": r[0], "ms": round(r[1], 3), "capacity_before": r[2], "capacity_after": r[3]}
                                                     for r in sorted(rec, key=lambda r: -r[1])[:12]]}}))
        return
    tm = Timer(torch, dist, dev, world)
    dt, dts, nxt = tm.run(step, barrier, args.steps, args.warmup)
    ms = dt / args.steps * 1e3
    n2 = min(max(6, args.steps), 60)
    gs.set_profiling(True); gd.set_profiling(True)
    acc = {}
    for i in range(n2):
        step(nxt + i)
        if i % 6 == 0:
            for k_, v_ in gs.counters().items():
                acc.setdefault(k_, []).append(v_)
            for k_, v_ in gd.counters().items():      # (the dynamic mapper's share of the pair launches' bytes)
                if k_ in ("tsdf_blocks_in_view", "esdf_columns_marked", "esdf_blocks_swept", "blocks_allocated"):
                    acc.setdefault("b_" + k_, []).append(v_)
    prof = gs.profile(); prof_d = gd.profile()
    gs.set_profiling(False); gd.set_profiling(False)
    for k_, v_ in prof_d.items():          # the dynamic (occupancy) mapper's launches join the static mapper's under the same kernel names
        if k_ == "_empty_event_pair":
            continue
        if k_ in prof:
            prof[k_] = {"count": prof[k_]["count"] + v_["count"], "total_ms": prof[k_]["total_ms"] + v_["total_ms"]}
        else:
            prof[k_] = v_
    counts = {k_: float(np.mean(v_)) for k_, v_ in acc.items()}
    # (the pair launches carry the static mapper's riders -- sphere tracing, candidates, marking of the held-back frame -- and its colour integration + transform:
    #  the formulas of the two-launch frame for the first mapper, the plain ones for the second; the classic kernels of a drain keep their own)
    pipelined_bytes = pair and not args.no_color_deferral
    kern, evo, emp = kernel_table(prof, counts, ms, n2, lambda k, cc: algorithmic_bytes(k, cc, rows, cols, fused=(pipelined_bytes and k.endswith("_pair"))), load_pmc("decay"))
    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        os_ = oracle.OracleMap(copy_params(oracle, gs.params)); od = oracle.OracleMap(copy_params(oracle, gd.params))
        tc = [0]

        def cstep(i):
            d, rgb, T = host[i % nu]
            mk = os_.detect_dynamics(d, T, cam, 5.0)
            mk = oracle.remove_small_components(mk, 2000)
            u_, m_ = oracle.split_depth_by_mask(d, mk, eye, cam, cam, 0.25)
            os_.set_time_ms(tc[0]); tc[0] += 33
            os_.integrate_depth(u_, T, cam); od.integrate_depth(m_, T, cam)
            os_.integrate_color(rgb, T, cam)
            os_.update_esdf(); od.update_esdf()
            if i % 6 == 5:
                os_.decay_tsdf(True); od.decay_occupancy()
        cstep(0); cstep(1)
        cpu = cpu_sample(lambda k: cstep(2 + k), args.cpu_seconds, 6, "frames/s", "frames of the same 640x480 dynamic sequence (all of the step)", oracle)
    # the timed sequence once more against the checker, outside the timed region: both maps emptied, the nu frames (decay on every 6th) mirrored on two
    # checker maps; static mapper: TSDF / colour / ESDF / slice, dynamic mapper: occupancy block set + log-odds, ESDF block set
    parity = None
    if not args.no_parity:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        ps_, pd_ = oracle.OracleMap(copy_params(oracle, gs.params)), oracle.OracleMap(copy_params(oracle, gd.params))
        gs.clear(); gd.clear(); barrier(); t_ms[0] = 0; tq = 0; t_par = time.perf_counter()
        occ_same = True; occ_worst = 0.0; occ_max = 0
        n_par = 2 * nu          # (two loops: the freespace layer needs the first to gain confidence, the second then meets dynamic pixels -- the occupancy mapper has blocks)
        for i in range(n_par):
            step(i)
            d, rgb, T = host[i % nu]
            mk = oracle.remove_small_components(ps_.detect_dynamics(d, T, cam, 5.0), 2000)
            u_, m_ = oracle.split_depth_by_mask(d, mk, eye, cam, cam, 0.25)
            ps_.set_time_ms(tq); tq += 33
            ps_.integrate_depth(u_, T, cam); pd_.integrate_depth(m_, T, cam); ps_.integrate_color(rgb, T, cam)
            ps_.update_esdf(); pd_.update_esdf()
            if i % 6 == 5:
                ps_.decay_tsdf(True); pd_.decay_occupancy()
            if i % 8 == 4:          # the dynamic (occupancy) mapper along the way: its layer comes and goes with the moving box
                io_, ig_ = pd_.block_indices(oracle.L_TSDF), gd.block_indices(M.LAYER_OCCUPANCY)
                same_ = bool(np.array_equal(ig_, io_)); occ_same = occ_same and same_; occ_max = max(occ_max, int(len(ig_)))
                if same_ and len(io_):
                    bg, _ = gd.get_blocks(M.LAYER_OCCUPANCY, ig_)
                    occ_worst = max(occ_worst, float(max(np.abs(bg[k_]["log_odds"].astype(np.float64) - pd_.get_block(oracle.L_TSDF, idx)["distance"]).max() for k_, idx in enumerate(io_))))
        barrier()
        parity = map_parity(M, gs, ps_, oracle)
        parity["dynamic_mapper"] = {"occupancy_blocks_max_along_the_way": occ_max, "index_sets_equal_every_8th_frame": occ_same, "max_abs_log_odds": occ_worst,
                                    "esdf_index_sets_equal": bool(np.array_equal(gd.block_indices(M.LAYER_ESDF), pd_.block_indices(oracle.L_ESDF)))}
        parity["ok"] = bool(parity["ok"] and occ_same and occ_worst <= 1e-4 and parity["dynamic_mapper"]["esdf_index_sets_equal"])
        parity["steps_compared"] = n_par; parity["checker_s"] = round(time.perf_counter() - t_par, 2)
        parity["what"] = ("the %d-frame sequence twice from two empty maps (front end, both mappers, colour, both ESDF updates, decay on every 6th frame), every call "
                          "mirrored on oracle/nvblox_oracle.c; outside the timed region" % nu)
    c = gs.counters()
    out = {"metric": "frames/s, dynamic mapping frame (dynamics + TSDF/freespace + occupancy + Color + ESDF, decay every 6th), synthetic Redwood-like 640x480 @0.05m",
           "value": round(args.steps / dt, 2), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "repeats": len(dts),
           "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[2]: synthetic Redwood-like room 8x6x2.8 m with a box moving at 0.5 m/s (SURVEY 8d), 640x480 depth+colour "
                                  "limited to 5 m, 0.05 m voxels, MappingType::kDynamic (freespace layer, dynamics detection, occupancy mapper), "
                                  "invalid_depth_decay 0.8, tsdf_decay 0.95 + occupancy decay every 6th frame", "unique_frames": nu,
                      "mode": (("colour deferral on both mappers (a new mapper's default, DESIGN.md 2.8), colour images in library-owned frames (retained, not copied); " if use_frames else
                                "colour deferral, staged form (raw device pointers: one k_stage_color copy per held-back frame), on both mappers; ") if not args.no_color_deferral else "classic launch order; ") +
                              ("the dynamic (occupancy) mapper on a stream of its own, ordered with nvbx_mapper_wait_for (A/B)" if args.own_stream else
                               "both mappers on one stream (as nvblox::MultiMapper hands them out)") +
                              ("; the two mappers' depth frames through nvbx_integrate_depth_pair (two launches for both, what MultiMapper::integrateDepth calls)" if pair else
                               "; the two mappers' depth frames as two nvbx_integrate_depth calls (four launches; --no-depth-pair)")},
           "depth_pair": pair,
           "readme_rtx5090_ms": README_RTX5090_MS,
           "per_step_counts": {k_: round(v_, 1) for k_, v_ in counts.items()},
           "block_ms": [round(d / args.steps * 1e3, 4) for d in dts[:16]], "block_stats_ms_per_step": block_stats(dts, args.steps),
           "kernels": kernels_json(kern),
           "roofline": roofline_of(kern, ms, evo, emp, "longest kernel of the dynamic-mapping step (launches of the static and the dynamic mapper together)"),
           "cpu_baseline": cpu, "parity": parity, "capacity_overflow": c["capacity_overflow"]}
    out["roofline"]["traffic_source"] = pmc_source("decay")
    print(json.dumps(out))



# ====================================================================================================== node cadence (the node's shape, not the fuser's)
def main_node(args):
    """One simulated second of NvbloxNode::tick() at the rates of nvblox_base.yaml:13-23: 40 depth frames, 5 colour frames, 10 updateEsdf each followed at
    once by the distance-slice query (processEsdf -> sliceAndPublishEsdf, nvblox_node.cpp:774-889: the slice is downloaded, so every ESDF tick drains the
    pipeline and waits), 5 updateColorMesh, 5 decayTsdfExcludeLastView, 1 clearOutsideRadius(7 m) -- in tick() order (depth, colour, ESDF, mesh, decay,
    clearing: nvblox_node.cpp:582-678) on the ticks where their periods coincide.  A block of `steps` = 40 depth slots is one simulated second; five seconds
    walk the 200-pose loop once.  Reported: GPU + host milliseconds per simulated second, launches per depth slot, drains per second, and per README tag the
    time a timing::Timer around the call reads as called (under deferral held-back work is not in it) and ATTRIBUTED (classic order, each call waited for)."""
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    torch, dist, rank, world, local_rank, dev = init_dist(args)
    assert world == 1, "the node-cadence line is single-GPU"
    cam = S.REPLICA_LIKE_CAM; rows, cols = cam[5], cam[4]
    scene = S.Scene(); nu = 200
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        T = S.trajectory_pose(i, 200); d, rgb = S.render(scene, T, cam); return (d, rgb, T)
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        host = list(pool.map(one, range(nu)))
    stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
    g = M.Mapper(M.default_params(), device=local_rank, block_capacity=1 << 14, stream=stream.cuda_stream)
    depth_dev = [torch.from_numpy(d).to(dev) for d, _, _ in host]
    rgb_frames = [M.ColorFrame(rows, cols, 3, local_rank).write(torch.from_numpy(c).to(dev), stream.cuda_stream) for _, c, _ in host]
    dargs = [g.prepare_depth(depth_dev[k], host[k][2], cam) for k in range(nu)]
    cargs = [g.prepare_color(rgb_frames[k], host[k][2], cam) for k in range(nu)]
    torch.cuda.synchronize(dev)
    SLOTS = 40; RADIUS = 7.0
    tags = ("tsdf/integrate", "color/integrate", "esdf/integrate", "esdf/slice", "mesh/integrate", "decay", "clear_outside_radius")
    acc = {t: [0.0, 0] for t in tags}
    timing_on = [False]; wait_each = [False]

    def call(tag, fn):
        if not timing_on[0]:
            return fn()
        t = time.perf_counter(); r = fn()
        if wait_each[0]:
            g.synchronize()
        a = acc[tag]; a[0] += (time.perf_counter() - t) * 1e3; a[1] += 1
        return r

    def slot(i):
        k = i % nu; s = i % SLOTS
        call("tsdf/integrate", lambda: g.integrate_prepared(dargs[k]))
        if s % 8 == 0:
            call("color/integrate", lambda: g.integrate_prepared(cargs[k]))
        if s % 4 == 0:
            call("esdf/integrate", g.update_esdf)
            call("esdf/slice", g.esdf_slice_image)            # -> host, as the node publishes it: drains and waits
        if s % 8 == 0:
            call("mesh/integrate", g.update_color_mesh)
            call("decay", lambda: g.decay_tsdf(True))
        if s == 0:
            T = host[k][2]
            call("clear_outside_radius", lambda: g.clear_outside_radius((float(T[0, 3]), float(T[1, 3]), float(T[2, 3])), RADIUS))

    def barrier():
        g.synchronize(); torch.cuda.synchronize(dev)

    def fresh():
        if fresh.pos == 0:
            g.clear()
        fresh.pos = (fresh.pos + SLOTS) % nu
    fresh.pos = 0
    tm = Timer(torch, dist, dev, world)
    dt, dts, base = tm.run(slot, barrier, SLOTS, SLOTS, before_block=fresh)
    ms_second = float(np.mean(dts)) * 1e3
    # per-tag host timers: as called (deferral as shipped), then attributed (classic order, every call waited for)
    def tag_pass(classic):
        for t in tags:
            acc[t] = [0.0, 0]
        g.set_color_deferral(not classic, staged=True); wait_each[0] = classic
        g.clear(); barrier(); timing_on[0] = True
        for i in range(nu):
            slot(i)
        barrier(); timing_on[0] = False; wait_each[0] = False
        return {t: {"ms_per_call": round(a[0] / max(1, a[1]), 4), "calls_per_second": a[1] / (nu / SLOTS), "ms_per_second": round(a[0] / (nu / SLOTS), 4)} for t, a in acc.items()}
    as_called = tag_pass(False)
    attributed = tag_pass(True)
    g.set_color_deferral(True, staged=True)
    # launches per slot / drains per second from the library's own spans
    g.clear(); barrier(); g.set_profiling(True)
    for i in range(nu):
        slot(i)
    prof = g.profile(); g.set_profiling(False); prof.pop("_empty_event_pair", None)
    launches = sum(v["count"] for v in prof.values())
    drains = sum(v["count"] for k_, v in prof.items() if short(k_).startswith("k_sphere_trace"))      # (a replay outside the pipeline launches the tracing on its own)
    kern = {short(k_): {"launches_per_second": round(v["count"] / (nu / SLOTS), 2), "ms_per_second": round(v["total_ms"] / (nu / SLOTS), 4)} for k_, v in prof.items()}
    # roofline of the longest kernel, per depth slot, with the counts of the loop's last state (the same formulas and PMC tables as the camera line)
    cnt_ = {k_: float(v_) for k_, v_ in g.counters().items()}
    ktab, evo_, emp_ = kernel_table(dict(prof), cnt_, ms_second / SLOTS, nu, lambda k_, cc: algorithmic_bytes(k_, cc, rows, cols), load_pmc("node"), exclude_from_calibration=())
    roofline = roofline_of(ktab, ms_second / SLOTS, evo_, emp_, "longest kernel per depth slot of the node-cadence second (launch counts per slot are fractional: colour, ESDF, mesh and decay "
                           "ticks are rarer than depth); durations = hipEvent spans minus the calibrated instrumentation cost", skip=())
    roofline["traffic_source"] = pmc_source("node")
    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        oc = oracle.OracleMap(copy_params(oracle, g.params))

        def cslot(i):
            k = i % nu; s_ = i % SLOTS; d, c_, T = host[k]
            oc.integrate_depth(d, T, cam)
            if s_ % 8 == 0:
                oc.integrate_color(c_, T, cam)
            if s_ % 4 == 0:
                oc.update_esdf(); oc.esdf_slice_image()
            if s_ % 8 == 0:
                oc.update_mesh(); oc.decay_tsdf(True)
            if s_ == 0:
                oc.clear_outside_radius((float(T[0, 3]), float(T[1, 3]), float(T[2, 3])), RADIUS)
        cpu = cpu_sample(cslot, args.cpu_seconds, SLOTS, "frames/s", "depth slots of the same node-cadence sequence (all of the tick's calls)", oracle)
    parity = None
    if not args.no_parity:
        import oracle
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        o = oracle.OracleMap(copy_params(oracle, g.params))
        g.clear()
        slices_equal = True
        for i in range(nu):
            k = i % nu; s = i % SLOTS; d, c_, T = host[k]
            slot(i)
            o.integrate_depth(d, T, cam)
            if s % 8 == 0:
                o.integrate_color(c_, T, cam)
            if s % 4 == 0:
                o.update_esdf()
                if s % 20 == 0:
                    sg, _ = g.esdf_slice_image(); so, _ = o.esdf_slice_image()
                    slices_equal = slices_equal and sg.shape == so.shape and float(np.abs(sg - so).max()) <= 1e-4
            if s % 8 == 0:
                o.update_mesh(); o.decay_tsdf(True)
            if s == 0:
                o.clear_outside_radius((float(T[0, 3]), float(T[1, 3]), float(T[2, 3])), RADIUS)
        g.update_esdf(); o.update_esdf()
        parity = map_parity(M, g, o, oracle)
        parity["slices_along_the_way_equal"] = bool(slices_equal); parity["ok"] = bool(parity["ok"] and slices_equal)
        parity["what"] = "the five simulated seconds once more, every call mirrored on oracle/nvblox_oracle.c; outside the timed region"
    out = {"metric": "depth frames/s at the node's call cadence (nvblox_base.yaml:13-23), synthetic Replica-like 640x480 @0.05m", "value": round(SLOTS / (ms_second * 1e-3), 2),
           "unit": "frames/s", "n_gpus": 1, "steps": SLOTS, "warmup": SLOTS, "repeats": len(dts), "ms_per_step": round(ms_second / SLOTS, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "node cadence: per simulated second 40 integrateDepth, 5 integrateColor, 10 updateEsdf + distance-slice download, 5 updateColorMesh, "
                                  "5 decayTsdfExcludeLastView, 1 clearOutsideRadius(7 m), in NvbloxNode::tick() order; 200-pose loop = 5 s, map emptied per loop",
                      "mode": "a new mapper's default (colour deferral, images in library frames)"},
           "ms_per_simulated_second": round(ms_second, 4), "gpu_utilisation_at_real_time": round(ms_second / 1000.0, 6),
           "launches_per_depth_slot": round(launches / nu, 3), "drains_per_second": round(drains / (nu / SLOTS), 2),
           "tags_as_called": as_called, "tags_attributed": attributed,
           "tags_note": "as_called: host timer around each call under the default deferral -- color/integrate and esdf/integrate read the ENQUEUE time only, their "
                        "kernels run inside the next tsdf/integrate or the next query (esdf/slice pays for the drain); attributed: classic launch order with a wait "
                        "after every call -- what the README's per-tag timers mean (README.md:69-97)",
           "kernels": kern, "roofline": roofline, "cpu_baseline": cpu, "readme_rtx5090_ms": README_RTX5090_MS, "parity": parity, "block_ms": [round(d * 1e3, 4) for d in dts[:16]]}
    print(json.dumps(out))
    finish_dist(dist, world)


# ====================================================================================================== camera / multicam
def main_camera(args):
    from isaac_ros_nvblox_amd import mapper as M, synthetic as S
    from isaac_ros_nvblox_amd.dist import PipelinedMeasurementFusion, PipelinedDirtyBlockExchange, camera_yaw_offset_deg
    torch, dist, rank, world, local_rank, dev = init_dist(args)
    multicam = args.workload == "multicam"
    ncam = max(1, min(8, args.cameras)) if multicam else 1
    assert not (multicam and world > 1), "multicam is the ONE-GPU form of configs[3]; N GPUs: --workload camera --gpus N"

    cam = S.REPLICA_LIKE_CAM
    rows, cols = cam[5], cam[4]
    # --scene hall: NOT the metric's configuration -- the same camera, trajectory and parameters in a 14 x 12 x 3 m hall, where the 8 m integration range
    # is used: 3-4 x the blocks in view of SURVEY 8d's 6 x 5 x 3 m room (whose ~300 are the low end of SURVEY 8a's own estimate for a real room)
    scene = S.Scene() if args.scene == "room" else S.Scene(room_min=(-7.0, -6.0, 0.0), room_max=(7.0, 6.0, 3.0))
    # distinct rendered frames, whatever --steps is: the whole 200-pose loop of SURVEY 8d for the metric's configuration (a block of K
    # steps integrates K CONSECUTIVE poses of it), 24 per camera for the 8-camera sweep
    nu = max(2, args.unique_frames)
    if multicam:
        nu = max(2, min(nu, 24))
    stride = max(1, 200 // nu)

    def render_cam(ci):
        from concurrent.futures import ThreadPoolExecutor
        yaw = camera_yaw_offset_deg(rank if not multicam else ci, 8)

        def one(i):
            T = S.trajectory_pose(i * stride, 200, yaw_offset_deg=yaw)
            d, rgb = S.render(scene, T, cam)
            return (d, rgb, T)
        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:      # (numpy releases the GIL: ~3x on 8 cores)
            return list(pool.map(one, range(nu)))
    host_cams = [render_cam(ci) for ci in range(8 if multicam else 1)]      # (multicam: all 8 rendered once, the sweep uses prefixes)
    depth_dev = [[torch.from_numpy(d).to(dev) for d, _, _ in fr] for fr in host_cams]
    rgb_dev = [[torch.from_numpy(c).to(dev) for _, c, _ in fr] for fr in host_cams]
    poses = [[T for _, _, T in fr] for fr in host_cams]
    host_frames = host_cams[0]

    stream = torch.cuda.Stream(dev)      # one explicit stream for torch ops, RCCL hand-off and every mapper kernel
    torch.cuda.set_stream(stream)
    g = M.Mapper(M.default_params(), device=local_rank, block_capacity=1 << 14, stream=stream.cuda_stream)      # (the room is ~1 650 blocks; pools grow on demand)
    fuse = world > 1 and args.fusion == "measurements"     # ONE fused map on every rank (exact, dist.MeasurementFusion) instead of replicas + index union
    ex = PipelinedDirtyBlockExchange(4096, dev) if (world > 1 and not fuse) else None      # one packed all-gather per frame, joined one frame later
    # buffers of 1024 records x 4112 B per rank; only max(count) rounded up to 64 records (~320 for ~300 blocks in view, 1.3 MB) goes
    # to the collective, one frame behind the measurement (dist.PipelinedMeasurementFusion)
    mf = PipelinedMeasurementFusion(1024, dev) if fuse else None
    mf_prev = [None]

    # Where the resident colour images live (the depth images and the raw colour tensors are torch allocations):
    #   "frames" (default): in library-owned device frames (nvbx_frame_acquire, include/nvblox_hip.h) -- what the device memory of an
    #     nvblox::Image<Color> IS in the facade, i.e. where the node's colour conversion writes (nvblox_node.cpp:1237-1263).  A mapper that holds
    #     integrateColor back retains such a frame instead of copying it: no k_stage_color launch.
    #   "staged" (--staged-deferral): raw device pointers, the mapper copies each held-back image into a frame of its own (round 4's default line)
    #   "zero_copy" (--zero-copy-deferral): raw device pointers under the caller's contract (nvbx_mapper_set_color_deferral(m, 1))
    cform = "zero_copy" if args.zero_copy_deferral else ("staged" if args.staged_deferral else "frames")
    rgb_frames = [[M.ColorFrame(rows, cols, 3, local_rank).write(t, stream.cuda_stream) for t in fr] for fr in rgb_dev]
    dargs = [[g.prepare_depth(depth_dev[ci][k], poses[ci][k], cam) for k in range(nu)] for ci in range(len(host_cams))]
    cargs_raw = [[g.prepare_color(rgb_dev[ci][k], poses[ci][k], cam) for k in range(nu)] for ci in range(len(host_cams))]
    cargs_frames = [[g.prepare_color(rgb_frames[ci][k], poses[ci][k], cam) for k in range(nu)] for ci in range(len(host_cams))]
    batch_ok = hasattr(g, "integrate_depth_batch")
    bd = {}; bc_raw = {}; bc_frames = {}
    if multicam and batch_ok:
        for n_ in (1, 2, 4, 8):
            bd[n_] = [g.prepare_depth_batch([depth_dev[ci][k] for ci in range(n_)], [poses[ci][k] for ci in range(n_)], cam) for k in range(nu)]
            bc_raw[n_] = [g.prepare_color_batch([rgb_dev[ci][k] for ci in range(n_)], [poses[ci][k] for ci in range(n_)], cam) for k in range(nu)]
            bc_frames[n_] = [g.prepare_color_batch([rgb_frames[ci][k] for ci in range(n_)], [poses[ci][k] for ci in range(n_)], cam) for k in range(nu)]
    csel = [cargs_frames if cform == "frames" else cargs_raw, bc_frames if cform == "frames" else bc_raw]      # (switched for the other forms' figures below)

    def step(i, mesh=False, exchange=True, n=None, batched=None):
        n = ncam if n is None else n
        k = i % nu
        use_batch = (multicam and batch_ok and n in bd) if batched is None else batched
        xg = ex if exchange else None            # rank-0-only passes after the timed region must not enter a collective
        if xg is not None:
            xg.before_depth(g)                   # the depth pass writes this frame's block indices into the exchange buffer itself
        if mf is not None and exchange:
            # frame i: measured now; frame i-1: its payload all-gather runs beside this measurement, then every camera's measurements are
            # applied in rank order and ITS colour + ESDF run (one fused map, one frame of latency)
            mf.begin(g, depth_dev[0][k], poses[0][k], cam)
            done = mf.finish_previous(g)
            kp, mf_prev[0] = mf_prev[0], k
            if done:
                g.integrate_prepared(csel[0][0][kp]); g.update_esdf()
            return
        elif use_batch:
            g.integrate_prepared_batch(bd[n][k])     # n cameras' depth frames: ONE view-marking launch + ONE TSDF-update launch
        else:
            for ci in range(n):
                g.integrate_prepared(dargs[ci][k])   # MultiMapper::integrateDepth
        if xg is not None:
            xg.start(g)                          # dirty block indices: export + async RCCL all-gather (needs only the depth pass)
            xg.finish_previous(g, deferred=True) # join the PREVIOUS frame's all-gather; its union step rides in the colour launch below
        if use_batch:
            g.integrate_prepared_batch(csel[1][n][k])
        else:
            for ci in range(n):
                g.integrate_prepared(csel[0][ci][k])   # MultiMapper::integrateColor (+ marking of own and peers' dirty blocks)
        g.update_esdf()                          # MultiMapper::updateEsdf
        if mesh:
            g.update_color_mesh()

    def barrier():
        if ex is not None:
            ex.drain(g)          # the all-gather still in flight is joined and applied inside the timed region
            g.set_view_export(None)
        if mf is not None and mf_prev[0] is not None:
            for _ in range(mf.drain(g)):     # the frame still on its way: gathered, applied, coloured, swept -- inside the timed region
                g.integrate_prepared(csel[0][0][mf_prev[0]]); g.update_esdf()
            mf_prev[0] = None
        g.synchronize()          # launches anything the mapper holds back (the EDT of the last updateEsdf) and waits for its stream
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    tm = Timer(torch, dist, dev, world)
    # Cross-frame pipelining (nvbx_mapper_set_color_deferral): integrateColor(i) / updateEsdf(i) are held back and carried out by
    # integrateDepth(i+1) in two launches per frame instead of four (view marking(i+1) || sphere tracing(i) || colour candidates(i) || ESDF marking(i),
    # then TSDF update(i+1) || colour integration(i) || distance transform(i)).  Same calls, same map;
    # the caller keeps the colour image valid until its next call (the bench's images are resident).  --no-color-deferral: the classic order.
    # (batches: the depth batch carries the held-back colour batch; N > 1 ranks with the index exchange take the pipeline too -- the union step
    #  of the peers' lists rides in the fused TSDF-update launch, dist.PipelinedDirtyBlockExchange rotates three buffer sets; the measurement
    #  exchange (--fusion measurements) applies whole frames one step late and keeps the classic order)
    deferral = (world == 1 or not fuse) and not args.no_color_deferral and ((not multicam) or (batch_ok and ncam in bd))
    # the form: STAGED (the default of a new mapper -- what a host gets that only swaps the library; one copy launch per colour frame) or, with
    # --zero-copy-deferral, the opt-in form without the copy (the caller keeps the colour image unchanged until its next call: the bench's are resident)
    staged = deferral and not args.zero_copy_deferral          # (frames / staged: the mapper's default setting, nvbx_mapper_set_color_deferral(m, 2))
    g.set_color_deferral(deferral, staged=staged)

    # EXPLORING (the headline): the map is EMPTIED at the start of every loop over the nu unique poses and the timed blocks of K steps tile the
    # loop (K = 200 = nu: one block per loop; the driver's K = 20: ten blocks per loop, the map emptied before every tenth), so every pose is
    # integrated exactly once per map and block allocation, hash inserts and first-touch of the pools are inside the timed region, as they are
    # when the reference fuses a sequence (the README's per-component timers run over a whole dataset).  The figure is the MEAN over complete
    # loops (timed seconds / steps: the early, allocation-heavy blocks count in proportion); the blocks that start from the empty map are also
    # quoted on their own (`first_block_of_loop`).  REVISIT (beside it): the same blocks of K steps on the fully allocated map after one
    # untimed loop -- the steady state of a robot that stays in a mapped room.
    loop_pos, tags = [0], []

    def fresh_map():
        if loop_pos[0] == 0 or loop_pos[0] + args.steps > nu:
            g.clear()
            mf_prev[0] = None
            loop_pos[0] = 0
        tags.append(loop_pos[0])
        loop_pos[0] += args.steps
    timed_step = (lambda i: step(i, mesh=True)) if args.with_mesh else step       # (--with-mesh: profiling passes that need k_mesh in the trace)
    dt, dts, base = tm.run(timed_step, barrier, args.steps, args.warmup, before_block=fresh_map)
    per_loop, starts, whole, kept = complete_loops(tags, dts, nu, args.steps)
    dt = float(np.sum(kept)) / len(kept)                 # mean block of the complete loops
    dt_first = float(np.median([dts[i] for i in starts]))
    ms_per_step = dt / args.steps * 1e3
    fps = world * ncam * args.steps / dt
    for i in range(nu):                      # one untimed loop: the map is complete
        step(base + i)
    dt_rev, dts_rev, base = tm.run(step, barrier, args.steps, 0, min_ms=MIN_TIMED_MS / 4, first=base + nu)
    ms_revisit = dt_rev / args.steps * 1e3
    ms_classic = None; ms_classic_exploring = None; ms_forms = {}
    if deferral and not args.profile_run:    # the same revisit blocks in the classic launch order, for the record
        g.set_color_deferral(False)
        dt_c, dts_c, base = tm.run(step, barrier, args.steps, 0, min_ms=MIN_TIMED_MS / 4, first=base)
        ms_classic = dt_c / args.steps * 1e3
        # ... and the EXPLORING figure in classic order: what a host that only swaps the library gets for the headline sequence
        loop_pos[0] = 0; n_tags0 = len(tags)
        dt_ce, dts_ce, base = tm.run(step, barrier, args.steps, 0, min_ms=MIN_TIMED_MS / 2, before_block=fresh_map, first=0)
        _, _, whole_c, kept_c = complete_loops(tags[n_tags0:], dts_ce, nu, args.steps)
        ms_classic_exploring = float(np.sum(kept_c)) / len(kept_c) / args.steps * 1e3
        # ... and in the OTHER forms of the deferral (revisit blocks): images in library frames / raw pointers staged / raw pointers zero-copy
        ms_forms = {}
        keep_sel = list(csel)
        for form_ in ("frames", "staged", "zero_copy"):
            if form_ == cform:
                continue
            csel[0], csel[1] = (cargs_frames, bc_frames) if form_ == "frames" else (cargs_raw, bc_raw)
            g.set_color_deferral(True, staged=form_ != "zero_copy")
            dt_s, dts_s, base = tm.run(step, barrier, args.steps, 0, min_ms=MIN_TIMED_MS / 4, first=base)
            ms_forms[form_] = dt_s / args.steps * 1e3
        csel[0], csel[1] = keep_sel
        g.set_color_deferral(True, staged=staged)

    if rank != 0:
        return finish_dist(dist, world)

    # ---- rank 0 extras (outside the timed region): per-component times, per-kernel roofline, CPU baseline
    def timed(fn, n, warm=8):
        for i in range(warm):            # (the first calls of a loop differ: e.g. the first mesh update after many frames without one meshes every block dirtied since)
            fn(i)
        g.synchronize(); torch.cuda.synchronize(dev); t = time.perf_counter()
        for i in range(n):
            fn(i)
        g.synchronize(); torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / n * 1e3

    n2 = min(args.steps, 100)
    n2c = 100                                # iterations of the per-component loops (whatever --steps is: 20 iterations between two synchronisations measured the synchronisation)
    comp = {}
    sweep = None
    if args.profile_run:
        # under rocprofv3 (tools/gpu_round.sh): nothing but the timed step is launched, so the trace's per-kernel averages are the step's
        # (the classic-order comparison, the per-component calls and the per-frame latency loop launch the same kernels in other forms)
        out = {"metric": "frames/s, TSDF+Color+ESDF integrate per frame, synthetic Replica-like 640x480 @0.05m", "value": round(fps, 2), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": len(dts), "ms_per_step": round(ms_per_step, 4),
               "ms_per_step_revisit": round(ms_revisit, 4), "profile_run": True, "color_deferral": {"enabled": bool(deferral)},
               "config": {"workload": args.workload, "cameras_per_gpu": ncam, "unique_frames": nu}}
        print(json.dumps(out))
        return finish_dist(dist, world)
    if not multicam:
        comp["tsdf"] = timed(lambda i: g.integrate_prepared(dargs[0][(base + i) % nu]), n2c)
        comp["color"] = timed(lambda i: g.integrate_prepared(csel[0][0][(base + i) % nu]), n2c)

        def esdf_only(i):
            g.integrate_prepared(dargs[0][(base + i) % nu]); g.update_esdf()
        comp["esdf"] = max(0.0, timed(esdf_only, n2c) - comp["tsdf"])

        def mesh_only(i):
            g.integrate_prepared(dargs[0][(base + i) % nu]); g.update_color_mesh()
        comp["mesh"] = max(0.0, timed(mesh_only, n2c) - comp["tsdf"])
        # the whole frame WITH a mesh update (configs[1] names Mesh): depth + colour + updateEsdf + updateColorMesh per frame -- the mesh reads the
        # colour layer, so it drains the pipeline every frame (five launches) -- and at the reference's cadence, every 8th frame (mesh 5 Hz against
        # depth 40 Hz, nvblox_base.yaml:13-23)
        comp["frame_with_mesh_every_frame"] = timed(lambda i: step(base + i, mesh=True, exchange=False), n2c)
        comp["frame_with_mesh_every_8th"] = timed(lambda i: step(base + i, mesh=(i % 8 == 7), exchange=False), n2c)
    else:
        # 1 / 2 / 4 / 8 cameras through one mapper: sequential calls vs one batched launch set
        sweep = {}
        for n_ in (1, 2, 4, 8):
            e = {"sequential_ms": round(timed(lambda i: step(base + i, n=n_, batched=False), n2c), 4)}
            if batch_ok:
                e["batched_ms"] = round(timed(lambda i: step(base + i, n=n_, batched=True), n2c), 4)
            sweep[str(n_)] = e

    # per-frame latency (SURVEY 8d timing protocol): every frame is waited for, so this is the latency a caller sees, not the
    # pipelined throughput of the timed region.
    # Two loops: WALL = host clock around the three calls + nvbx_synchronize, nothing else in between (round 4 recorded two torch events, flushed and
    # synchronised the whole device inside the clocked span: ~8 us of the instrument's own); GPU = event pair around the same calls, in a loop of its own.
    lat_wall = []
    torch.cuda.synchronize(dev)
    for i in range(n2):
        t = time.perf_counter()
        step(base + i, exchange=False)
        g.synchronize()                      # replays what the frame holds back (colour, ESDF marking + distance transform) and waits for the mapper's stream
        lat_wall.append((time.perf_counter() - t) * 1e3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n2)]
    for i in range(n2):
        ev[i][0].record(stream)
        step(base + n2 + i, exchange=False)
        g.flush()                            # enqueue the held-back work of this frame
        ev[i][1].record(stream)
        g.synchronize(); torch.cuda.synchronize(dev)
    lat_gpu = [a.elapsed_time(b) for a, b in ev]
    pct = lambda v: {"mean": round(float(np.mean(v)), 4), "p50": round(float(np.percentile(v, 50)), 4), "p99": round(float(np.percentile(v, 99)), 4)}
    latency = {"frames": n2, "wall_ms": pct(lat_wall), "gpu_ms": pct(lat_gpu),
               "note": "one step at a time, waited for: wall = host clock around integrateDepth + integrateColor + updateEsdf + nvbx_synchronize (which replays the "
                       "held-back colour / ESDF work: four dependent launches per frame) -- the figure comparable to the README's host timers; gpu = an event "
                       "pair around the same calls, measured in a loop of its own"}

    # per-step work counts (a query flushes the pipeline, so they are sampled in a loop of their own, not in the profiled one)
    counts_acc = {}
    for i in range(min(n2, 20)):
        step(base + i, mesh=True, exchange=False)
        for k_, v_ in g.counters().items():
            counts_acc.setdefault(k_, []).append(v_)
    counts = {k_: float(np.mean(v_)) for k_, v_ in counts_acc.items()}
    # per-kernel durations with hipEvent pairs on the mapper stream: the timed step as it is (n2 frames, nothing in between), then a few
    # frames with a mesh update each for k_mesh's entry
    g.set_profiling(True)
    for i in range(n2):
        step(base + i, exchange=False)
    g.synchronize()
    prof = g.profile()
    g.set_profiling(True)
    for i in range(10):
        step(base + i, mesh=True, exchange=False)
    prof_mesh = g.profile()
    g.set_profiling(False)
    for k_, v_ in prof_mesh.items():
        if short(k_).startswith("k_mesh"):
            prof[k_] = {"count": v_["count"] * n2 / 10.0, "total_ms": v_["total_ms"] * n2 / 10.0}      # (scaled to one launch per step)
    launches_cam = ncam if not (multicam and batch_ok and ncam in bd) else 1
    n_trace_launches = sum(v_["count"] for k_, v_ in prof.items() if short(k_).startswith("k_sphere_trace"))
    fused_trace = deferral and n_trace_launches < n2 / 2          # (the last frame of the loop is flushed in classic order by the synchronize)
    fused_colc = sum(v_["count"] for k_, v_ in prof.items() if short(k_).startswith("k_integrate_tsdf_color")) > n2 / 2
    launches_per_frame = 2 if fused_colc else (3 if fused_trace else 4)
    if fused_trace:
        for k_ in [k_ for k_ in prof if short(k_).startswith("k_sphere_trace")]:
            del prof[k_]
    # the PMC table of THIS shape of the workload (the 4- and 8-camera lines carried one table): multicam = 4 cameras, multicam8 = 8; none for other counts
    pmc_key = ("multicam8" if ncam == 8 else "multicam" if ncam == 4 else "none") if multicam else ("camera_mesh" if args.with_mesh else "camera")
    if args.scene != "room":
        pmc_key = "none"          # (the committed PMC tables are the room's: no `traffic` for another scene)
    kern, ev_overhead_us, empty_pair_us = kernel_table(
        prof, counts, ms_revisit, n2, lambda k, cc: algorithmic_bytes(k, cc, rows, cols, n_cam=(ncam // launches_cam), trace_in_mark_view=fused_trace, fused=fused_colc), load_pmc(pmc_key),
        exclude_from_calibration=("k_mesh", "k_esdf_edt"))
    roofline = roofline_of(
        kern, ms_per_step, ev_overhead_us, empty_pair_us,
        "kernel = the LONGEST non-mesh kernel of the step by time (two launches per frame, DESIGN.md 2.8: k_mark_view = view marking of this frame + "
        "sphere tracing, colour candidates and ESDF marking of the held-back frame; k_integrate_tsdf_color = TSDF update + that frame's colour "
        "integration and distance transform; classic order, DESIGN.md 2.4: k_integrate_color also carries the ESDF site marking, k_mark_view the "
        "held-back EDT of the previous update); durations = span of a hipEvent pair around each launch on the mapper stream minus "
        "`event_pair_overhead_us`, the instrumentation cost per launch calibrated so that the step's launches add up to the un-instrumented "
        "step time; compare rocprofv3's kernel-trace averages in profiles/*_kernel_stats.csv.  640x480 @ 0.05 m moves ~10 MB per camera frame, "
        "so every kernel is bound by its dependent-access chain and launch cost rather than by HBM bytes (DESIGN.md 2)")

    # bytes the step moves that are NOT in SURVEY 8d's formulas (the staged form's copy of raw-pointer colour images): reported, never counted as algorithmic
    roofline["step"]["overhead_bytes"] = int(sum(overhead_bytes(k_, rows, cols, n_cam=(ncam // launches_cam)) * v_["launches_per_step"] for k_, v_ in kern.items()))
    roofline["traffic_source"] = pmc_source(pmc_key) if pmc_key != "none" else None

    cpu = None
    if not args.no_cpu_baseline:
        import oracle
        # 8 OpenMP threads: the oracle's parallel regions are one frame's blocks (a few hundred tasks), which stop scaling
        # there (1 thread 7.6 ms, 8 threads 3.4 ms, 256 threads 620 ms per frame on the EPYC 9575F host -- BASELINE.md 3)
        oracle.set_num_threads(min(8, os.cpu_count() or 1))
        o = oracle.OracleMap(copy_params(oracle, g.params))

        def cstep(k):
            for ci in range(ncam):
                d, c_, T = host_cams[ci][k % nu]; o.integrate_depth(d, T, cam)
            for ci in range(ncam):
                d, c_, T = host_cams[ci][k % nu]; o.integrate_color(c_, T, cam)
            o.update_esdf()
        cstep(0); cstep(1)
        cpu = cpu_sample(lambda k: cstep(2 + k), args.cpu_seconds, max(2, args.cpu_frames // ncam), "steps/s" if multicam else "frames/s",
                         "steps of the same 640x480 sequence (%d camera%s per step), TSDF+Color+ESDF" % (ncam, "" if ncam == 1 else "s"), oracle)
        if multicam:
            cpu["value"] = round(cpu["value"] * ncam, 3); cpu["unit"] = "frames/s"
        else:
            # BASELINE.json configs[0]: "TSDF+ESDF only, CPU reference path, single process" -- the same frames without colour, one thread and eight
            def c0(threads):
                oracle.set_num_threads(min(threads, os.cpu_count() or 1))
                o0 = oracle.OracleMap(copy_params(oracle, g.params))

                def s0(k):
                    d, c_, T = host_cams[0][k % nu]; o0.integrate_depth(d, T, cam); o0.update_esdf()
                s0(0); s0(1)
                r = cpu_sample(lambda k: s0(2 + k), max(2.0, args.cpu_seconds / 4), 8, "frames/s", "frames of the same 640x480 sequence, TSDF+ESDF only (configs[0])", oracle)
                return {k_: r[k_] for k_ in ("value", "unit", "cores", "ms_per_step", "sample")}
            cpu["configs0_tsdf_esdf_only"] = {"one_thread": c0(1), "eight_threads": c0(8)}
            oracle.set_num_threads(min(8, os.cpu_count() or 1))

    # the timed sequence once more, outside the timed region, against the checker (the 200-pose exploring loop with clear(),
    # deferral and a drain per block had never been compared end to end)
    parity = None
    if not args.no_parity:
        import oracle

        def checker_step(o, k):
            for ci in range(ncam):
                d, c_, T = host_cams[ci][k % nu]; o.integrate_depth(d, T, cam)
            for ci in range(ncam):
                d, c_, T = host_cams[ci][k % nu]; o.integrate_color(c_, T, cam)
            o.update_esdf()

        def drain_local():
            g.synchronize(); torch.cuda.synchronize(dev)
        parity = exploring_parity(M, g, lambda k: step(k, exchange=False), drain_local, nu, checker_step, args.steps, oracle)
        parity["mode"] = "color_deferral" if deferral else "classic"

    out = {
        "metric": "frames/s, TSDF+Color+ESDF integrate per frame, synthetic Replica-like 640x480 @0.05m",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": len(dts),
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[3] on one GPU: %d cameras (45 deg yaw offsets) through one mapper, %s; " % (ncam, "one batched launch set per step" if (batch_ok and ncam in bd) else "sequential calls") if multicam else "configs[1]: ") +
                               ("synthetic Replica-like room (SURVEY 8d)" if args.scene == "room" else "NOT the metric's scene -- a 14 x 12 x 3 m hall (--scene hall: 3-4 x the blocks in view), same camera and trajectory") +
                               ", 640x480 depth+colour, 0.05 m voxels, fuser.yaml params, TSDF+Color+ESDF every frame (mesh timed separately)",
                   "scene": args.scene,
                   "cameras_per_gpu": ncam, "parallelism": ("one camera per GPU, RCCL all-gather of per-voxel measurements, one fused map on every rank" if fuse else "one camera per GPU, RCCL all-gather of dirty block indices") if world > 1 else "single GPU",
                   "unique_frames": nu,
                   "mode": (("color_deferral, images in library-owned frames: a new mapper's default setting (nvbx_mapper_set_color_deferral(m, 2)) fed colour images that "
                             "live in frames of nvbx_frame_acquire -- the device memory of an nvblox::Image<Color> in the facade, where the node's conversion writes; the "
                             "mapper retains the frame of a held-back image instead of copying it (csrc/frames.hip).  Raw device pointers under the same setting are copied "
                             "first (color_deferral.ms_per_step_revisit_staged_copy, --staged-deferral); classic order of four launches: ms_per_step_classic_order") if cform == "frames" else
                            ("color_deferral, staged: nvbx_mapper_set_color_deferral(m, 2), the default of a new mapper, fed RAW device pointers (--staged-deferral): "
                             "each held-back image is copied into a frame of the mapper's own first (k_stage_color)") if staged else
                            ("color_deferral, zero-copy: nvbx_mapper_set_color_deferral(m, 1) -- OPT-IN (--zero-copy-deferral; the caller keeps its colour image "
                             "unchanged until its next call); the default of a new mapper is the staged form, color_deferral.ms_per_step_revisit_staged_copy")) if deferral
                           else "classic launch order (--no-color-deferral / NVBX_COLOR_DEFERRAL=0; a new mapper defaults to staged colour deferral)"},
        "ms_per_frame": round(ms_per_step / ncam, 4),
        "ms_per_step_classic_order": (round(ms_classic_exploring, 4) if ms_classic_exploring else (None if deferral else round(ms_per_step, 4))),
        "timing": {"value_is": "exploring: the map is emptied at the start of every loop over the %d unique poses; timed blocks of %d steps tile the loop "
                               "(%d per loop), every pose is integrated once per map (allocation inside the timed region); mean over the %d complete "
                               "loops timed" % (nu, args.steps, per_loop, len(whole)),
                   "loops": len(whole), "blocks_per_loop": per_loop,
                   "first_block_of_loop_ms_per_step": round(dt_first / args.steps * 1e3, 4),
                   "exploring_ms_per_step": block_stats(kept, args.steps),
                   "revisit_ms_per_step": block_stats(dts_rev, args.steps),
                   "revisit_note": "same blocks of K steps on the fully allocated map (after one untimed loop over all poses)"},
        "ms_per_step_revisit": round(ms_revisit, 4),
        "color_deferral": {"enabled": bool(deferral), "form": cform if deferral else None, "launches_per_frame": launches_per_frame,
                           "ms_per_step_revisit_classic_order": (round(ms_classic, 4) if ms_classic else None),
                           "ms_per_step_revisit_frames": (round(ms_forms["frames"], 4) if "frames" in ms_forms else None),
                           "ms_per_step_revisit_staged_copy": (round(ms_forms["staged"], 4) if "staged" in ms_forms else None),
                           "ms_per_step_revisit_zero_copy": (round(ms_forms["zero_copy"], 4) if "zero_copy" in ms_forms else None),
                           "frame_pool": dict(zip(("held", "free", "bytes", "created", "waits", "syncs"), M.frame_pool_stats())),
                           "note": "enabled: integrateColor(i) / updateEsdf(i) are held back and carried out by integrateDepth(i+1) in pipelined order: "
                                   "launch 1 = view marking(i+1) || sphere tracing(i) || colour candidates(i) || ESDF marking(i), launch 2 = TSDF update(i+1) "
                                   "|| colour integration(i) || distance transform(i) (NVBX_FUSE_COLC=0: three launches); same calls, bit-identical map "
                                   "(tests/test_gpu_pipeline.py); contract: include/nvblox_hip.h nvbx_mapper_set_color_deferral.  N > 1 GPUs with the index "
                                   "exchange run the same two launches: the union step of the peers' gathered lists rides in launch 2 (three rotating buffer "
                                   "sets, DESIGN.md 6.1); --fusion measurements keeps the classic order"},
        "block_ms": [round(d / args.steps * 1e3, 4) for d in dts[:16]],
        "ms_components": {k_: round(v_, 4) for k_, v_ in comp.items()},
        "components_note": "ms_components are measured per call in isolation: the EDT of updateEsdf is held back and runs inside the next depth "
                           "frame's first launch, so `esdf` shows the marking launch only and the sum of the three is not ms_per_step",
        "frame_latency": latency,
        "readme_rtx5090_ms": README_RTX5090_MS,
        "speedup_vs_readme_rtx5090_tsdf_color_esdf": round(0.7 / (ms_per_step / ncam), 2),
        "speedup_note": "indicative only: the README figures are host timers on the real Replica scene; the comparable figure here is frame_latency.wall_ms",
        "per_step_counts": {k_: round(v_, 1) for k_, v_ in counts.items()},
        "kernels": kernels_json(kern),
        "roofline": roofline,
        "cpu_baseline": cpu,
        "parity": parity,
    }
    if sweep is not None:
        out["camera_sweep_ms_per_step"] = sweep
    print(json.dumps(out))
    finish_dist(dist, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--unique-frames", type=int, default=200, help="distinct rendered frames cycled through (HBM-resident); independent of --steps")
    ap.add_argument("--cpu-frames", type=int, default=48, help="minimum number of frames of the same workload timed on the CPU oracle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU baseline keeps integrating (cycling the same frames) until this much CPU wall time has passed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="camera / multicam: skip the end-state comparison of the timed sequence with the checker (outside the timed region)")
    ap.add_argument("--own-stream", action="store_true", help="decay: the dynamic mapper of the dynamic-mapping frame on a stream of its own (A/B; slower: EXPERIMENTS.md)")
    ap.add_argument("--no-depth-pair", action="store_true", help="decay workload: the static and the dynamic mapper's depth frames as two calls (four launches) instead of nvbx_integrate_depth_pair (A/B)")
    ap.add_argument("--profile-run", action="store_true", help="camera / multicam: only the timed step is launched (for rocprofv3 runs: clean per-kernel averages)")
    ap.add_argument("--with-mesh", action="store_true", help="camera: the timed step also updates the colour mesh (TSDF+Color+ESDF+Mesh per frame; profiling passes for k_mesh)")
    ap.add_argument("--staged-deferral", action="store_true", help="camera workload: colour images as RAW device pointers under the default (staged) deferral: one k_stage_color copy per held-back frame (round 4's default line); default now: the images live in library-owned frames (nvbx_frame_acquire), no copy")
    ap.add_argument("--zero-copy-deferral", action="store_true", help="camera workload: colour deferral WITHOUT the staging copy (opt-in form: the caller keeps the colour image unchanged); default: staged, as a new mapper")
    ap.add_argument("--no-color-deferral", action="store_true", help="camera workload: classic launch order (4 launches per frame) instead of the cross-frame pipeline")
    ap.add_argument("--separate-front-end", action="store_true", help="decay workload: detect / remove-small-components / split as three entry points (A/B against nvbx_dynamic_depth_split)")
    ap.add_argument("--step-trace", type=int, default=0, help="decay workload: wait for every one of this many steps and report the slowest (diagnosis)")
    ap.add_argument("--cameras", type=int, default=4, help="multicam: cameras per step (1..8)")
    ap.add_argument("--scene", default="room", choices=["room", "hall"], help="camera / multicam: room = SURVEY 8d's 6 x 5 x 3 m room (the metric's configuration); hall = 14 x 12 x 3 m, 3-4 x the blocks in view (a scaling check, not the metric)")
    ap.add_argument("--fusion", default="indices", choices=["indices", "measurements"],
                    help="N > 1 GPUs: indices = replicas + all-gather of dirty block indices (north-star wording, default); "
                         "measurements = all-gather of per-voxel measurements, ONE fused map on every rank (SURVEY 8e option B, exact)")
    ap.add_argument("--workload", default="camera", choices=["camera", "multicam", "decay", "lidar", "node"],
                    help="camera = BASELINE.json configs[1] (the metric's configuration, default); multicam = configs[3] on one GPU; "
                         "decay = configs[2]; lidar = configs[4]")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU) -- the same command line the driver uses
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.workload == "lidar":
        return main_lidar(args)
    if args.workload == "node":
        args.steps = 40
        return main_node(args)
    if args.workload == "decay":
        return main_decay(args)
    return main_camera(args)


if __name__ == "__main__":
    main()
