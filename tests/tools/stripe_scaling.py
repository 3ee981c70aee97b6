"""Experiment tool (GPU box): what one rank of an N-GPU job has to do, timed on one GPU -- the interleaved row
blocks of rank 0 and rank N-1 of N for N = 1, 2, 4, 8.  The ratio t(1) / max over ranks t(N) BOUNDS the strong-scaling
speed-up a node can show (the gather of the film stripes comes on top).  It is a bound measured on one device, not a
multi-GPU measurement.

    python3 tests/tools/stripe_scaling.py [config] [R] [N,N,...] [--json FILE]
        config: c1 (default: teapot_scene 640x640x100), c4 (shiny_teapot 1920x1080x512), c5 (bust 2048x2048x1024),
                c5_64 (bust at 64 spp: the same per-round structure at 1/16 of the time)
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hobbyraytracer_amd import api  # noqa: E402

CONFIGS = {"c1": ("teapot_scene.yaml", 640, 640, 100, 5), "c4": ("shiny_teapot.yaml", 1920, 1080, 512, 3),
           "c5": ("bust_scene.yaml", 2048, 2048, 1024, 1), "c5_64": ("bust_scene.yaml", 2048, 2048, 64, 2)}
args = [a for a in sys.argv[1:]]
json_out = None
if "--json" in args:
    i = args.index("--json"); json_out = args[i + 1]; del args[i:i + 2]
cfg = args[0] if args and args[0] in CONFIGS else "c1"
if args and args[0] in CONFIGS:
    args = args[1:]
R = int(args[0]) if args else 8
Ns = [int(x) for x in args[1].split(",")] if len(args) > 1 else [1, 2, 4, 8]
scene, W, H, spp, reps = CONFIGS[cfg]
tmp = tempfile.mkdtemp(prefix="hrt_scal_")
api.write_teapot_obj(os.path.join(tmp, "teapot.obj"), 1.0)
api.write_hall_hdr(os.path.join(tmp, "old_hall_4k.hdr"), 4096, 2048)
if "bust" in scene:
    api.write_bust_obj(os.path.join(tmp, "marble_bust_01.obj"), 1.0)
hs = api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", scene), tmp)
dev = api.DeviceScene(hs.flat_ptr, 0)
cam = hs.camera(W, H)
p = api.default_params(W, H, spp, timing=True)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t1 = None
rows = []
for G in Ns:
    worst = 0.0
    for rank in sorted({0, G - 1}):
        dev.render_stripes_device(cam, p, R, rank, G, out.data_ptr(), stream)
        torch.cuda.synchronize()
        dev.stats()
        for _ in range(reps):
            dev.render_stripes_device(cam, p, R, rank, G, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = dev.stats()
        ms = st.kernel_ms / st.launches
        worst = max(worst, ms)
        if t1 is None:
            t1 = ms
        print(f"{cfg} N={G} rank {rank}: {ms:.2f} ms per frame share, speed-up bound {t1 / ms:.2f}x, {st.rays / reps / ms / 1e3:.0f} Mrays/s", flush=True)
    gather_mb = api.stripe_rows(H, R, 0, G) * W * 12 / 1e6
    rows.append({"n_gpus": G, "share_ms": round(worst, 3), "speedup_bound": round(t1 / worst, 3), "gather_mb_per_rank": round(gather_mb, 2)})
if json_out:
    json.dump({"config": cfg, "scene": scene, "film": [W, H], "spp": spp, "rows_per_block": R,
               "what": "one rank's share of an N-GPU frame timed on ONE MI355X (slower of rank 0 and rank N-1): a bound on strong scaling, not a multi-GPU measurement",
               "rows": rows}, open(json_out, "w"), indent=1)
