#!/bin/bash
# usage (GPU box, repo root): tests/tools/bench_configs.sh <tag>  -- the BASELINE.json configurations that fit one MI355X,
# one bench line each into gpurun_out/cfg_<tag>/; the table in DESIGN.md section 6 is made from these files.
set -o pipefail
out=gpurun_out/cfg_$1
mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-cli-wall-clock "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }
        python3 -c "import json,sys; b=json.load(open('$out/$name.json')); r=b['roofline']; print('%-14s %8.0f Mrays/s %8.1f ms  ext %7.1f ms frac %s' % ('$name', b['value'], b['ms_per_step'], r.get('kernel_ms_per_frame') or 0, r.get('frac')))"; }
run c1
run c1_fixed --quirks fixed
run c1_mega --megakernel --steps 3 --warmup 1
run c2 --scene cornell_box.yaml --spp 256
run c3 --scene teapot_scene.yaml --width 1024 --height 1024 --spp 256 --quirks fixed --steps 3 --warmup 1
run c4 --scene shiny_teapot.yaml --width 1920 --height 1080 --spp 512 --steps 3 --warmup 1
run c5_64 --scene bust_scene.yaml --width 2048 --height 2048 --spp 64 --steps 3 --warmup 1
run c5 --scene bust_scene.yaml --width 2048 --height 2048 --spp 1024 --steps 1 --warmup 0
