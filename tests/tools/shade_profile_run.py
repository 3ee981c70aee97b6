"""Wave-cycles of k_wf_shade by phase (GPU box, repo root).  Needs a library built with -DHRT_SHADE_PROFILE (s_memtime at the phase
boundaries of wf_shade_task, summed per wave):
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DHRT_SHADE_PROFILE -shared -o build_variants/libhrt_hip_sprof.so \\
          hobbyraytracer_amd/csrc/hrt_hip.hip -Iinclude -ldl
    cp build_variants/libhrt_hip_sprof.so hobbyraytracer_amd/lib/libhrt_hip.so && python3 tests/tools/shade_profile_run.py
The library prints one "[shade profile]" line per hrt_scene_stats call (stderr).  Round 2, headline frame (640x640x100, 50 rounds): state
load (+ trailing prims: none in this scene) 26 % of a wave's cycles -- latency, hidden by the other three waves of the SIMD --,
hitRecord + scatter 45 %, next-segment preparation 25 %, miss queue 1 %, stores + enqueue 3 %."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from hobbyraytracer_amd import api
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 256, 128)
hs = api.HostScene("tests/golden/scenes/teapot_scene.yaml", d)
dev = api.DeviceScene(hs.flat_ptr, 0)
W = H = 640
cam = hs.camera(W, H)
for md in (1, 2, 50):
    p = api.default_params(W, H, 100, max_depth=md)
    sys.stderr.write(f"--- max_depth {md}\n"); sys.stderr.flush()
    img, st = dev.render_tile(cam, p)
    sys.stderr.write(f"    rays {st.rays} kernel_ms {st.kernel_ms:.2f}\n")
