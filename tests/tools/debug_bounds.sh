#!/bin/bash
# tests/tools/debug_bounds.sh [pytest args]   (GPU box, from the repo root)
# Builds libhrt_hip.so with -DHRT_DEBUG_BOUNDS into build_variants/bounds/ and runs the GPU suite on it: every table index a
# hit record is built from (triangle of a mesh, prim, material, texture, mesh, frontFace source) is checked on the device;
# tests/conftest.py fails a test that ends with a non-zero violation counter.  Run ONCE after a change to the sub-index flags
# (HRT_SUB_*), the state records or the hit-record path -- not a standing part of the suite (the kernels are ~3 % slower).
set -eo pipefail
out=build_variants/bounds
mkdir -p $out
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -DHRT_DEBUG_BOUNDS"
if [ ! -f $out/libhrt_hip.so ] || [ hobbyraytracer_amd/csrc/hrt_hip.hip -nt $out/libhrt_hip.so ] || [ hobbyraytracer_amd/csrc/hrt_device.h -nt $out/libhrt_hip.so ]; then
    $HIPCC $FLAGS -c -o $out/hrt_hip.o hobbyraytracer_amd/csrc/hrt_hip.hip
    $HIPCC $FLAGS -c -o $out/hrt_lbvh.o hobbyraytracer_amd/csrc/hrt_lbvh.hip
    $HIPCC $FLAGS -c -o $out/hrt_sahbvh.o hobbyraytracer_amd/csrc/hrt_sahbvh.hip
    $HIPCC $FLAGS -shared -o $out/libhrt_hip.so $out/hrt_hip.o $out/hrt_lbvh.o $out/hrt_sahbvh.o -ldl
fi
export HRT_HIP_LIB=$PWD/$out/libhrt_hip.so
export LD_LIBRARY_PATH=$PWD/$out:$LD_LIBRARY_PATH      # the CLI binary too (RUNPATH is searched after LD_LIBRARY_PATH)
python3 - <<PY
from hobbyraytracer_amd import api
v = api.debug_bounds_violations(0)
assert v is not None, "not a -DHRT_DEBUG_BOUNDS build"
print("debug-bounds build loaded:", api.HIP_LIB_PATH)
PY
python3 -m pytest tests -m gpu -x -q "$@"
