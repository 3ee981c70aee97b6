#!/bin/bash
# tests/tools/ta_probe.sh   (GPU box, repo root)
# Is k_wf_ext bound by the L1's address / tag pipeline, or by vector issue?  (also -DHRT_VALU_PROBE=k: k more v_fma per node step)
# Is k_wf_ext bound by the L1's address / tag pipeline?  Builds libhrt_hip.so with -DHRT_TA_PROBE=k (k = 1, 2: k more 16-byte loads of
# the node record per node step -- same cache line, results unused, no longer dependency chain) and times the headline frame's
# kernels with each against the product build on the same box.
set -eo pipefail
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value"
for k in 1 2; do
    out=build_variants/ta$k; mkdir -p $out
    $HIPCC $FLAGS -DHRT_TA_PROBE=$k -c -o $out/hrt_hip.o hobbyraytracer_amd/csrc/hrt_hip.hip
    $HIPCC $FLAGS -shared -o $out/libhrt_hip.so $out/hrt_hip.o build/hrt_lbvh.o build/hrt_sahbvh.o -ldl
done
for k in 14 28; do      # + 25 % / + 50 % of a node step's ~55 vector instructions
    out=build_variants/valu$k; mkdir -p $out
    $HIPCC $FLAGS -DHRT_VALU_PROBE=$k -c -o $out/hrt_hip.o hobbyraytracer_amd/csrc/hrt_hip.hip
    $HIPCC $FLAGS -shared -o $out/libhrt_hip.so $out/hrt_hip.o build/hrt_lbvh.o build/hrt_sahbvh.o -ldl
done
bash tests/tools/kernel_times.sh ta_base | grep -E "k_wf_ext|sum"
for k in 1 2; do HRT_HIP_LIB=$PWD/build_variants/ta$k/libhrt_hip.so bash tests/tools/kernel_times.sh ta_probe$k | grep -E "k_wf_ext|sum"; done
for k in 14 28; do HRT_HIP_LIB=$PWD/build_variants/valu$k/libhrt_hip.so bash tests/tools/kernel_times.sh ta_valu$k | grep -E "k_wf_ext|sum"; done
bash tests/tools/kernel_times.sh ta_base2 | grep -E "k_wf_ext|sum"
rm -rf gpurun_out/kt_ta_*/
