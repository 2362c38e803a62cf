"""Compiler-reported resource usage of every kernel (no GPU needed): VGPRs, SGPRs, spills, scratch, LDS, occupancy.
Usage: python tools/kernel_resources.py > profiles/<tag>_kernel_resources.md"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "isaac_ros_nvblox_amd", "csrc")
rows = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            short = re.sub(r"\(.*", "", dem).replace("void ", "").replace("nvbx::", "")
            cur = {"file": os.path.basename(src), "kernel": short}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
print("| file | kernel | VGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | LDS B/workgroup | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|---|")
for c in rows:
    print("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (c["file"], c["kernel"], c.get("VGPRs"), c.get("TotalSGPRs"), c.get("SGPRs Spill"), c.get("VGPRs Spill"),
                                                             c.get("ScratchSize [bytes/lane]"), c.get("LDS Size [bytes/block]"), c.get("Occupancy [waves/SIMD]")))
