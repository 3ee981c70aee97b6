import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from hobbyraytracer_amd import api
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 256, 128)
hs = api.HostScene("tests/golden/scenes/teapot_scene.yaml", d)
dev = api.DeviceScene(hs.flat_ptr, 0)
for (W, H, spp) in ((64, 64, 4), (128, 128, 16), (256, 256, 32)):
    cam = hs.camera(W, H); p = api.default_params(W, H, spp)
    dev.render_tile(cam, p)
    t0 = time.time()
    for _ in range(20): dev.render_tile(cam, p)
    print("%dx%dx%d: %.2f ms" % (W, H, spp, (time.time() - t0) / 20 * 1e3))
