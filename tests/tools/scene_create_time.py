import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from hobbyraytracer_amd import api
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_bust_obj(d + "/marble_bust_01.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 4096, 2048)
for scene in ("teapot_scene.yaml", "bust_scene.yaml"):
    t0 = time.perf_counter(); hs = api.HostScene("tests/golden/scenes/" + scene, d); t1 = time.perf_counter()
    dev = api.DeviceScene(hs.flat_ptr, 0); dev.close()
    ts = []
    for _ in range(3):
        t2 = time.perf_counter(); dev = api.DeviceScene(hs.flat_ptr, 0); ts.append(time.perf_counter() - t2); dev.close()
    print(f"{scene}: load {1e3*(t1-t0):.1f} ms, hrt_scene_create {1e3*min(ts):.1f} ms (nodes {hs.flat.n_nodes}, tris {hs.flat.n_tris})")
