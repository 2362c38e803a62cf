#!/usr/bin/env python3
"""Digest rocprofv3 output (gpurun_out/...) into the tracked summaries under profiles/.

usage: summarize_profile.py TAG STATS_DIR [PMC_FETCH_DIR PMC_WRITE_DIR] [--workload W]
  - copies *_kernel_stats.csv to profiles/TAG_kernel_stats.csv (W != camera: profiles/TAG_W_kernel_stats.csv)
  - per kernel: mean FETCH_SIZE / WRITE_SIZE per launch -> HBM bytes per launch, with the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): units are KiB, FETCH_SIZE counts 64 B per 128-B request (x2) -- which the guide
    calibrates for wide coalesced reads and tools/pmc_calibrate.sh confirms for THIS library's patterns (8-byte voxels per lane, 64 contiguous
    bytes per lane, scattered 16-byte hash entries and 8-byte voxels: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, i.e. a scattered access costs a
    whole 128-B line); WRITE_SIZE is exact (64-B requests).  Written to profiles/TAG[_W]_pmc.json and merged into profiles/pmc_latest.json under the workload's
    key (read by bench.py `traffic`).
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name


def counter_means(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def main():
    argv = list(sys.argv[1:])
    workload = "camera"
    if "--workload" in argv:
        i = argv.index("--workload"); workload = argv[i + 1]; del argv[i:i + 2]
    tag, stats_dir = argv[0], argv[1]
    suffix = "" if workload == "camera" else "_" + workload
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s%s_kernel_stats.csv" % (tag, suffix)))
    if len(argv) >= 4:
        fetch = counter_means(argv[2], "FETCH_SIZE")
        write = counter_means(argv[3], "WRITE_SIZE")
        out = {}
        for k in sorted(set(fetch) | set(write)):
            if not k.startswith("k_"):
                continue
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            out[k] = {"FETCH_SIZE_KiB": round(fk, 3), "WRITE_SIZE_KiB": round(wk, 3),
                      "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
                      "note": "2x FETCH_SIZE gfx950 correction, 1x WRITE_SIZE: calibrated on known byte counts in this library's access patterns (profiles/r06_pmc_calibration.json: every read request is 128 B, counted at 64)"}
        json.dump(out, open(os.path.join(ROOT, "profiles", "%s%s_pmc.json" % (tag, suffix)), "w"), indent=1)
        latest_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        try:
            latest = json.load(open(latest_path))
        except Exception:
            latest = {}
        if any(k.startswith("k_") or k.startswith("__amd") for k in latest):      # round-1 layout (flat = camera)
            latest = {"camera": {k: v for k, v in latest.items() if k.startswith("k_")}}
        latest[workload] = out
        # where each workload's table comes from: bench.py prints it as roofline.traffic_source (the counters are never measured inside a bench run)
        import subprocess
        try:
            commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
        except Exception:
            commit = "unknown"
        meta = latest.get("_meta") or {}
        meta.setdefault("by_workload", {})[workload] = {"tag": tag, "commit_when_summarised": commit}
        meta["tag"] = tag; meta["commit"] = commit
        latest["_meta"] = meta
        json.dump(latest, open(latest_path, "w"), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
