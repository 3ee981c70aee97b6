#!/bin/bash
# usage: pmc.sh <tag> "<COUNTER COUNTER ...>" [bench args...]
# One rocprofv3 --pmc pass (own run, --kernel-trace only) over one bench.py frame; prints the counters summed over
# the frame per wavefront kernel and merges them into gpurun_out/pmc_<tag>.json.  (GPU box)
tag=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
d=gpurun_out/pmc_${tag}_$(echo $ctrs | cksum | cut -d' ' -f1)
rm -rf $d
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli-wall-clock "$@" > $d.log 2>&1
python3 - <<PY
import csv,glob,json,os,collections,re
fs=glob.glob("$d/*/*_counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "<true" in n or not ("k_wf" in n or "k_pathtrace" in n): continue
    k=re.search(r"(k_wf_[a-z]+|k_pathtrace)", n).group(1)
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    calls[(k,r["Counter_Name"])]+=1
out="gpurun_out/pmc_$tag.json"
old=json.load(open(out)) if os.path.exists(out) else {}
for k,v in acc.items():
    old.setdefault(k,{}).update({c:{"sum_over_one_frame":x,"launches":calls[(k,c)]} for c,x in v.items()})
    print(k," ".join("%s=%.4g"%(c,x) for c,x in sorted(v.items())))
json.dump(old,open(out,"w"),indent=1,sort_keys=True)
PY
