"""usage (GPU box, repo root): python3 tests/tools/gpu_fuzz.py FIRST LAST -- a fuzz campaign, not a test: random worlds
(tests/scene_helpers.py random_world; mild and extreme, with and without meshes) on every render path against the oracle.
HRT_FUZZ_LBVH=1: with the meshes' culling trees built on the GPU (a load that fails because the LBVH is deeper than 31 levels is skipped).
Failing worlds are copied to gpurun_out/gfuzz/.  About 40 worlds per second."""
import sys, os, tempfile, pathlib, time; sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api
from oracle import oracle_py as orc
from tests.scene_helpers import random_world
d = pathlib.Path(tempfile.mkdtemp())
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if os.environ.get("HRT_FUZZ_LBVH"):           # the meshes' culling trees from the GPU builder (hrt_bvh_build_device) instead of the host's
    api.use_device_bvh_builder(True)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)
bad = 0; ran = 0; t0 = time.time()
for extreme in (0, 1):
    for meshes in (True, False):
        for seed in range(lo, hi):
            path = random_world(d, 5000 + seed, extreme, meshes=meshes, images=(seed % 2 == 1), bare_glass=(seed % 3 == 0))
            os.dup2(devnull, 1)
            msg = None
            try:
                try:
                    hs = api.HostScene(path, str(d))
                except api.HrtError:
                    if os.environ.get("HRT_FUZZ_LBVH"): continue
                    raise
                try:
                    dev = api.DeviceScene(hs.flat_ptr, 0)
                except api.HrtError as e:
                    continue
                world = orc.World(hs.flat_ptr)
                cam = hs.camera(40, 40)
                ran += 1
                for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
                    ref, sr = world.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=True))
                    for tail, mega, stats in (("1", False, True), ("1000", False, False), ("3", False, True), ("1", True, False)):
                        os.environ["HRT_WF_TAIL_ROUND"] = tail
                        img, st = dev.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=stats, megakernel=mega))
                        nd = int(((img.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(img) & np.isnan(ref))).any(2).sum())   # NaN == NaN here: payloads may differ, Film::tonemap scrubs both
                        if nd or st.rays != sr.rays:
                            msg = (seed, extreme, meshes, q, tail, mega, stats, nd, st.rays, sr.rays)
                dev.close()
            finally:
                os.dup2(saved, 1)
            if msg:
                bad += 1; print("DIFF", msg, flush=True)
                os.makedirs("gpurun_out/gfuzz", exist_ok=True)
                import shutil; shutil.copy(path, "gpurun_out/gfuzz/fail_%d_%d_%d.yaml" % (seed, extreme, int(meshes)))
    print("progress extreme", extreme, "ran", ran, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
print("done ran", ran, "bad", bad)
