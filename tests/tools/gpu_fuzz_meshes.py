"""usage (GPU box, repo root): python3 tests/tools/gpu_fuzz_meshes.py FIRST LAST [quirks] -- a fuzz campaign, not a test: structured
scenes (tests/scene_helpers.py many_meshes_scene: 1..10 instances of a teapot / a quad, glass and metal spheres, a floor; random
placements) at 96x96x8 through the C ABI against the oracle, quirks=fixed by default.  About 9 M path segments per second
including the oracle's share on the box's 16 host threads."""
import os, sys, tempfile, pathlib, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api
from oracle import oracle_py as orc
from tests.scene_helpers import many_meshes_scene, films_equal

first, last = int(sys.argv[1]), int(sys.argv[2])
q = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = pathlib.Path(tempfile.mkdtemp())
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)
bad = n = rays = 0
t0 = time.time()
W = H = 96; spp = 8
for seed in range(first, last):
    n_mesh = 1 + seed % 10
    os.dup2(devnull, 1)
    try:
        hs = api.HostScene(many_meshes_scene(d, n_mesh, seed=5000 + seed), str(d))
        dev, world, cam = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr), hs.camera(W, H)
        ref, sr = world.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True))
        img, st = dev.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True))
        ok = films_equal(img, ref) and st.rays == sr.rays
        nd = int(((img.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(img) & np.isnan(ref))).any(2).sum())
        dev.close()
    finally:
        os.dup2(saved, 1)
    n += 1; rays += sr.rays
    if not ok:
        bad += 1; print("DIFF seed", seed, "meshes", n_mesh, "px", nd, "rays", st.rays - sr.rays, flush=True)
    if n % 500 == 0:
        print("progress", n, "bad", bad, "segments %.2f G" % (rays / 1e9), "%.0fs" % (time.time() - t0), flush=True)
print("done", n, "films, bad", bad, ", segments %.2f G" % (rays / 1e9))
