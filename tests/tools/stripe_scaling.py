"""Experiment tool (GPU box): what one rank of an N-GPU job has to do, timed on one GPU -- the interleaved row
blocks of rank 0 of N for N = 1, 2, 4, 8 (bench.py's workload).  The ratio t(1) / t(N) bounds the strong-scaling
speed-up the driver can measure (the all_gather of <= 0.6 MB per rank comes on top)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hobbyraytracer_amd import api  # noqa: E402

W, H, spp, R = 640, 640, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 8
tmp = tempfile.mkdtemp(prefix="hrt_scal_")
api.write_teapot_obj(os.path.join(tmp, "teapot.obj"), 1.0)
api.write_hall_hdr(os.path.join(tmp, "old_hall_4k.hdr"), 4096, 2048)
hs = api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", "teapot_scene.yaml"), tmp)
dev = api.DeviceScene(hs.flat_ptr, 0)
cam = hs.camera(W, H)
p = api.default_params(W, H, spp, timing=True)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t1 = None
for G in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    for rank in (0, G - 1):
        dev.render_stripes_device(cam, p, R, rank, G, out.data_ptr(), stream)
        torch.cuda.synchronize()
        dev.stats()
        for _ in range(5):
            dev.render_stripes_device(cam, p, R, rank, G, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = dev.stats()
        ms = st.kernel_ms / st.launches
        if t1 is None:
            t1 = ms
        print(f"N={G} rank {rank}: {ms:.2f} ms per frame share, speed-up bound {t1 / ms:.2f}x, {st.rays / 5 / ms / 1e3:.0f} Mrays/s", flush=True)
