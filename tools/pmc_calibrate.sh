#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts in this library's access patterns (tools/micro/pmc_calibrate.hip).  usage: tools/pmc_calibrate.sh TAG
TAG=${1:-cal}; R=$PWD; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/pmc_calibrate.hip -o /tmp/pmc_calibrate || exit 1
/tmp/pmc_calibrate > gpurun_out/$TAG/moved.json
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -o pmc -- /tmp/pmc_calibrate > /dev/null 2> $R/gpurun_out/$TAG/fetch.err); echo "fetch rc=$?"
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -o pmc -- /tmp/pmc_calibrate > /dev/null 2> $R/gpurun_out/$TAG/write.err); echo "write rc=$?"
(cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/gpurun_out/$TAG/rdreq -o pmc -- /tmp/pmc_calibrate > /dev/null 2> $R/gpurun_out/$TAG/rdreq.err); echo "rdreq rc=$?"
(cd /tmp && rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum --output-format csv -d $R/gpurun_out/$TAG/wrreq -o pmc -- /tmp/pmc_calibrate > /dev/null 2> $R/gpurun_out/$TAG/wrreq.err); echo "wrreq rc=$?"
python - <<PY
import csv, glob, json, re
moved = json.load(open("gpurun_out/$TAG/moved.json"))
def counter(d, name):
    out = {}
    for f in glob.glob("gpurun_out/$TAG/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            m = re.search(r"(cal_[a-z_0-9]+)", r["Kernel_Name"])
            if m: out[m.group(1)] = out.get(m.group(1), 0.0) + float(r["Counter_Value"])
    return out
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
rd = {c: counter("rdreq", c) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
wr = {c: counter("wrreq", c) for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_BUBBLE_sum")}
res = {"what": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB as reported) of ONE launch per pattern over a 2 GiB buffer, beside the bytes the launch moved; "
               "factor = bytes moved / (counter x 1024): what the counter has to be multiplied by", "patterns": {}}
for k, mv in moved.items():
    if not isinstance(mv, dict): continue
    row = {"moved": mv, "FETCH_SIZE_KiB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KiB": round(write.get(k, 0.0), 1)}
    fb, wb = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    for name, b in mv.items():
        if name.startswith("read") and fb > 0: row["fetch_factor_vs_" + name] = round(b / fb, 3)
        if name.startswith("write") and wb > 0: row["write_factor_vs_" + name] = round(b / wb, 3)
    row["requests"] = {c.replace("TCC_EA0_", "").replace("_sum", ""): rd[c].get(k, 0.0) for c in rd}
    row["requests"].update({c.replace("TCC_EA0_", "").replace("_sum", ""): wr[c].get(k, 0.0) for c in wr})
    q = row["requests"]
    row["read_bytes_by_request_size"] = int(32 * q["RDREQ_32B"] + 64 * q["RDREQ_64B"] + 128 * q["RDREQ_128B"])
    res["patterns"][k] = row
    print(k, {a: b for a, b in row.items() if "factor" in a or "KiB" in a}, row["requests"], row["read_bytes_by_request_size"])
json.dump(res, open("gpurun_out/$TAG/pmc_calibration.json", "w"), indent=1)
PY
