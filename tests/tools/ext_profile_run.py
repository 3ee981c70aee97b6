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
