"""usage (GPU box, repo root): rocprofv3 --kernel-trace --stats -d gpurun_out/bvhprof -- python3 tests/tools/bvh_build_profile.py [lbvh|sah] [teapot|bust]
Calls the GPU BVH builder ten times on one mesh and prints the wall-clock per call (the kernel times come from rocprofv3)."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api
algo = sys.argv[1] if len(sys.argv) > 1 else "sah"
mesh = sys.argv[2] if len(sys.argv) > 2 else "bust"
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_bust_obj(d + "/marble_bust_01.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 64, 32)
hs = api.HostScene("tests/golden/scenes/" + ("bust_scene.yaml" if mesh == "bust" else "teapot_scene.yaml"), d)
pos = np.asarray(hs.mesh_arrays(0)[0], dtype=np.float32).reshape(-1, 9)
api.bvh_build_device(pos, 2, algo=algo)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); api.bvh_build_device(pos, 2, algo=algo); ts.append(1e3 * (time.perf_counter() - t0))
print(f"{algo} {mesh}: {pos.shape[0]} triangles, wall per build: min {min(ts):.2f} ms, median {sorted(ts)[5]:.2f} ms")
