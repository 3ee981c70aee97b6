#!/bin/bash
# usage: kernel_times.sh <tag> [bench args...]   -- rocprofv3 per-kernel totals of one bench.py run (GPU box)
tag=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli-wall-clock "$@" > gpurun_out/kt_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kt_$tag/*/*_kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_wf" in n or "k_pathtrace" in n:
        ms=float(r["TotalDurationNs"])/1e6
        if "<true" not in n:   # 1 warmup + 3 timed frames run these kernels
            import re; short=re.search(r"(k_wf_[a-z]+|k_pathtrace)", n).group(1)
            print("  %-22s calls %5s  %8.2f ms/frame"%(short, r["Calls"], ms/4.0)); tot+=ms/4.0
print("$tag: sum %.2f ms/frame"%tot)
PY
