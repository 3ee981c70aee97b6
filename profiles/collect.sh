#!/bin/bash
# profiles/collect.sh <name>   (run on the MI355X box from the repo root, e.g. through gpurun)
# Produces gpurun_out/<name>/{bench_line.json,kernel_stats.csv,pmc_per_kernel.json,traffic.json}: the default
# bench line, the rocprofv3 --kernel-trace --stats summary of the same command, and the PMC passes (each in its
# own rocprofv3 run with --kernel-trace only, <= 8 SQ / 4 TCC counters per pass) summed over one frame per kernel.
# Copy the directory to profiles/<name>/ and commit it.
set -o pipefail
name=${1:-r01_new}
out=gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp

timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
echo "bench: $(cut -c1-200 $out/bench_line.json)"

rm -rf $out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cli-wall-clock > $out/kt.log 2>&1 || { echo "kernel-trace failed"; exit 1; }
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
echo "kernel stats done"

for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
    timeout -k 10 150 tests/tools/pmc.sh $name "$group" > /dev/null || { echo "pmc pass failed: $group"; exit 1; }
    echo "pmc pass done: $group"
done
cp gpurun_out/pmc_$name.json $out/pmc_per_kernel.json

python3 - <<PY
import json
d = json.load(open("$out/pmc_per_kernel.json"))
k = d.get("k_wf_ext", {})
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    n = k["FETCH_SIZE"]["launches"]
    fetch = k["FETCH_SIZE"]["sum_over_one_frame"] * 1024 / n
    write = k["WRITE_SIZE"]["sum_over_one_frame"] * 1024 / n
    json.dump({"kernel": "k_wf_ext", "launches_in_profile": n, "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
               "hbm_bytes_per_launch": 2 * fetch + write,
               "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches: FETCH_SIZE under-reports wide reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM)"},
              open("$out/traffic.json", "w"), indent=1)
    print("traffic: %.1f MB per k_wf_ext launch" % ((2 * fetch + write) / 1e6))
PY
rm -rf $out/kt $out/kt.log gpurun_out/pmc_${name}_* gpurun_out/pmc_$name.json
ls $out
