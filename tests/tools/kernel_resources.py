"""Prints hipcc's resource report (VGPRs, scratch, LDS, occupancy) of every kernel in csrc/hrt_hip.hip (no GPU needed).
  python3 tests/tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
       "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "hobbyraytracer_amd", "csrc", "hrt_hip.hip"), "-o", "/tmp/_kr.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
name = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); rows[name] = {}
        continue
    m = re.search(r"remark:\s+(SGPRs|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and name:
        rows[name][m.group(1).split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-34s VGPR %s SGPR %s scratch %s LDS %s occ %s" % (short, v.get("VGPRs"), v.get("SGPRs"), v.get("ScratchSize"), v.get("LDS"), v.get("Occupancy")))
