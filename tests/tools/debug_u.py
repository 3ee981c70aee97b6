import sys, os, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hobbyraytracer_amd import api
from oracle import oracle_py as orc
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 512, 256)
img = np.zeros((32, 64, 3), np.uint8); img[:, ::8] = [255, 40, 40]; api.write_image(d + "/stripes.png", img)
hs = api.HostScene("tests/golden/scenes/material_zoo.yaml", d)
dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
r = np.random.default_rng(17)
o = r.uniform([-3, 0.05, -3], [3, 4.5, 7], (150000, 3)).astype(np.float32)
dd = (r.uniform([-2.5, 0, -2.5], [2.5, 3.5, 2.5], (150000, 3)) - o).astype(np.float32)
p = api.default_params(16, 16, 1, seed=5)
g, c = dev.closest_hit(p, o, dd, pixel0=1000), world.closest_hit(p, o, dd, pixel0=1000)
hit = c["prim"] >= 0
for f in ("t", "u", "v"):
    bad = hit & (g[f].view(np.uint32) != c[f].view(np.uint32))
    print(f, "mismatches", bad.sum(), "by prim", np.unique(c["prim"][bad], return_counts=True))
    idx = np.nonzero(bad)[0][:8]
    for i in idx:
        print("   prim", c["prim"][i], "kind", hs.flat.prims[int(c["prim"][i])].kind, "gpu", g[f][i], "cpu", c[f][i], "ulps", int(g[f][i].view(np.uint32)) - int(c[f][i].view(np.uint32)), "normal", c["normal"][i], "p", c["p"][i])
