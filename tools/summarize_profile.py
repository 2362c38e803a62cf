#!/usr/bin/env python3
"""Digest rocprofv3 output (gpurun_out/...) into the tracked summaries under profiles/.

usage: summarize_profile.py TAG STATS_DIR [PMC_FETCH_DIR PMC_WRITE_DIR]
  - copies *_kernel_stats.csv to profiles/TAG_kernel_stats.csv
  - per kernel: mean FETCH_SIZE / WRITE_SIZE per launch -> HBM bytes per launch, with the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): units are KiB, FETCH_SIZE counts 64 B per 128-B request on wide
    coalesced reads (x2).  Written to profiles/TAG_pmc.json and profiles/pmc_latest.json (read by bench.py `traffic`).
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
    tag, stats_dir = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    if len(sys.argv) >= 5:
        fetch = counter_means(sys.argv[3], "FETCH_SIZE")
        write = counter_means(sys.argv[4], "WRITE_SIZE")
        out = {}
        for k in sorted(set(fetch) | set(write)):
            fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
            out[k] = {"FETCH_SIZE_KiB": round(fk, 3), "WRITE_SIZE_KiB": round(wk, 3),
                      "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
                      "note": "2x FETCH_SIZE gfx950 correction (calibrated for 16-B/lane streams; our 8-B/lane accesses are uncalibrated)"}
        for name in ("%s_pmc.json" % tag, "pmc_latest.json"):
            json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
