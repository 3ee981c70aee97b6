"""Where the 64 lanes of k_wf_ext are, phase by phase (GPU box, repo root).  Needs a library built with -DHRT_EXT_PROFILE:
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DHRT_EXT_PROFILE -shared -o /tmp/libhrt_hip_prof.so \\
          hobbyraytracer_amd/csrc/hrt_hip.hip -ldl
    cp /tmp/libhrt_hip_prof.so hobbyraytracer_amd/lib/libhrt_hip.so && python3 tests/tools/ext_profile_run.py
The library prints one "[ext profile]" line per hrt_scene_stats call (stderr).  Round 2, headline frame (640x640x100, 50 rounds):
10.5 M outer iterations, 107.7 M inner wave-steps with 41 % of the 64 lanes AT an inner node while 87 % hold a ray (the rest wait at
a postponed leaf or are done), 10.2 M leaf phases with 50 % of the lanes at a leaf (90 % of those leaves hold two triangles).
With -DHRT_STEP_PROFILE instead (s_memtime stamps inside a node step, after forced waits; lane 0 speaks for its wave; the build spills 60
bytes, so the parts are what counts, not the total): "[step profile]" -- 2060 ticks per node step of a wave = 1299 until the data of the two
16-byte node loads is there, 333 box tests, 255 descend / push / pop until the LDS answers, 173 votes and loop control."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from hobbyraytracer_amd import api
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 256, 128)
hs = api.HostScene("tests/golden/scenes/teapot_scene.yaml", d)
dev = api.DeviceScene(hs.flat_ptr, 0)
W = H = 640
cam = hs.camera(W, H)
for md in (2, 3, 6, 12, 25, 50):
    p = api.default_params(W, H, 100, max_depth=md)
    sys.stderr.write(f"--- max_depth {md}\n"); sys.stderr.flush()
    img, st = dev.render_tile(cam, p)
    sys.stderr.write(f"    rays {st.rays} kernel_ms {st.kernel_ms:.2f}\n")
