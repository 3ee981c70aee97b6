"""usage (GPU box, repo root): python3 tests/tools/gpu_fuzz_launch.py N [seed] -- a fuzz campaign over the LAUNCH bookkeeping of
the wavefront pipeline, not a test: random film sizes, tiles, stripes, sample ranges, depths and the experiment knobs (batch
size, task size, tail round, dynamic task pulling, blocks per CU, leaf threshold) on three scenes; every result must be
bit-identical to the megakernel's (which the parity suites pin to the oracle)."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api

N = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.random.default_rng(seed)
d = tempfile.mkdtemp()
api.write_teapot_obj(os.path.join(d, "teapot.obj"), 0.5)
api.write_hall_hdr(os.path.join(d, "old_hall_4k.hdr"), 256, 128)
api.write_bust_obj(os.path.join(d, "marble_bust_01.obj"), 0.1)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)
os.dup2(devnull, 1)
scenes = {}
for name in ("teapot_scene", "three_meshes", "material_zoo", "cornell_box"):
    try:
        hs = api.HostScene(f"tests/golden/scenes/{name}.yaml", d)
        scenes[name] = (hs, api.DeviceScene(hs.flat_ptr, 0))
    except Exception as e:
        os.dup2(saved, 1); print("skip", name, str(e)[:100]); os.dup2(devnull, 1)
os.dup2(saved, 1)
KNOBS = {"HRT_WF_MAX_SLOTS": [None, 64, 200, 1000, 4096, 50000], "HRT_WF_TASK_SIZE": [None, 64, 128, 256, 1024, 4096],
         "HRT_WF_TAIL_ROUND": [None, 1, 2, 3, 5, 9, 1000], "HRT_WF_DYNAMIC_TASKS": [None, 1], "HRT_EXT_BLOCKS_PER_CU": [None, 1, 2, 6, 10],
         "HRT_EXT_LEAF_NUM": [None, 1, 16, 48, 64], "HRT_WF_TASK_GROUPS": [None, 1, 7, 64, 191, 256]}
bad = 0
for it in range(N):
    name = list(scenes)[int(r.integers(0, len(scenes)))]
    hs, dev = scenes[name]
    W, H = int(r.choice([1, 2, 3, 7, 16, 33, 64, 97])), int(r.choice([1, 2, 5, 8, 17, 40, 64]))
    W, H = max(W, 2), max(H, 2)
    spp = int(r.choice([1, 2, 3, 4, 7, 16]))
    md = int(r.choice([1, 2, 3, 7, 50]))
    q = int(r.choice([api.QUIRKS_REFERENCE, api.QUIRKS_FIXED]))
    stats = bool(r.integers(0, 2))
    env = {k: v[int(r.integers(0, len(v)))] for k, v in KNOBS.items()}
    for k, v in env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    cam = hs.camera(W, H)
    pm = api.default_params(W, H, spp, quirks=q, stats=True, megakernel=True, max_depth=md)
    pw = api.default_params(W, H, spp, quirks=q, stats=stats, max_depth=md)
    mode = int(r.integers(0, 3))
    try:
        if mode == 0:      # a tile
            x0, y0 = int(r.integers(0, W)), int(r.integers(0, H))
            rect = (x0, y0, int(r.integers(1, W - x0 + 1)), int(r.integers(1, H - y0 + 1)))
            a, sa = dev.render_tile(cam, pm, rect); b, sb = dev.render_tile(cam, pw, rect)
            what = ("tile", rect)
        elif mode == 1:    # stripes of one rank
            R, G = int(r.choice([1, 2, 8, 16])), int(r.choice([1, 2, 3, 8]))
            g = int(r.integers(0, G))
            if api.stripe_rows(H, R, g, G) == 0: continue
            a, sa = dev.render_stripes(cam, pm, R, g, G); b, sb = dev.render_stripes(cam, pw, R, g, G)
            what = ("stripes", R, g, G)
        else:              # progressive: random split of the samples, must end equal to the one-shot stripes
            R, G = int(r.choice([1, 8])), int(r.choice([1, 2]))
            g = int(r.integers(0, G))
            rows = api.stripe_rows(H, R, g, G)
            if rows == 0: continue
            a, sa = dev.render_stripes(cam, pm, R, g, G)
            b = np.zeros((rows, W, 3), np.float32)
            s0 = 0; rays = 0
            while s0 < spp:
                n = int(r.integers(1, spp - s0 + 1))
                st = dev.render_stripes_accumulate(cam, pw, R, g, G, b, s0, n); rays += st.rays; s0 += n
            sb = st; sb.rays = rays
            what = ("progressive", R, g, G)
    except api.HrtError as e:
        print("ERROR", it, name, W, H, spp, md, q, env, str(e)[:200], flush=True); bad += 1; continue
    nd = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    if nd or sa.rays != sb.rays:
        bad += 1
        print("DIFF", it, name, W, H, spp, "depth", md, "quirks", q, "stats", stats, what, env, "px", nd, "rays", sa.rays, sb.rays, flush=True)
    if it % 200 == 199: print("progress", it + 1, "bad", bad, flush=True)
print("done", N, "bad", bad)
