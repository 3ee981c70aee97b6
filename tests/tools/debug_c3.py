import sys, os, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hobbyraytracer_amd import api
from oracle import oracle_py as orc
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 512, 256)
hs = api.HostScene("tests/golden/scenes/teapot_scene.yaml", d)
dev = api.DeviceScene(hs.flat_ptr, 0)
W = H = 1024; spp = 256
cam = hs.camera(W, H)
x0, y0 = W // 2 - 16, H // 2
rect = (x0, y0, 32, 8)
pw = api.default_params(W, H, spp, quirks=api.QUIRKS_FIXED)
pm = api.default_params(W, H, spp, quirks=api.QUIRKS_FIXED, megakernel=True)
ref, sr = orc.World(hs.flat_ptr).render_tile(cam, pw, rect)
tw, sw = dev.render_tile(cam, pw, rect)
tm, sm = dev.render_tile(cam, pm, rect)
print("tile wavefront == oracle", np.array_equal(tw.view(np.uint32), ref.view(np.uint32)), "rays", sw.rays, sr.rays)
print("tile megakernel == oracle", np.array_equal(tm.view(np.uint32), ref.view(np.uint32)), "rays", sm.rays)
full, sf = dev.render_tile(cam, pw)
blk = full[y0:y0 + 8, x0:x0 + 32]
print("full wavefront block == oracle", np.array_equal(blk.view(np.uint32), ref.view(np.uint32)))
bad = np.argwhere((blk.view(np.uint32) != ref.view(np.uint32)).any(2))
print("differing pixels in block", len(bad), bad[:5].tolist())
for (r, c) in bad[:3]:
    print("  px", r, c, "full", blk[r, c], "oracle", ref[r, c], "tile", tw[r, c])
os.environ["HRT_WF_MAX_SLOTS"] = str(W * H * 300)
full1, _ = dev.render_tile(cam, pw)   # one batch (268 M slots = 49 GB workspace)
print("full one-batch block == oracle", np.array_equal(full1[y0:y0 + 8, x0:x0 + 32].view(np.uint32), ref.view(np.uint32)), "== chunked", np.array_equal(full1.view(np.uint32), full.view(np.uint32)))
