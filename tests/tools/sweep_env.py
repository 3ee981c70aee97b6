"""Experiment tool (GPU box): frame / traversal time for several values of one launch-time environment knob
of libhrt_hip.so (read on every render call), e.g.
    python3 tests/tools/sweep_env.py HRT_EXT_LEAF_NUM 32,40,48,56 [scene.yaml W H spp quirks]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hobbyraytracer_amd import api  # noqa: E402

knob, values = sys.argv[1], sys.argv[2].split(",")
scene = sys.argv[3] if len(sys.argv) > 3 else "teapot_scene.yaml"
W, H, spp = (int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (640, 640, 100)
quirks = api.QUIRKS_FIXED if (len(sys.argv) > 7 and sys.argv[7] == "fixed") else api.QUIRKS_REFERENCE
tmp = tempfile.mkdtemp(prefix="hrt_sweep_")
api.write_teapot_obj(os.path.join(tmp, "teapot.obj"), 1.0)
api.write_hall_hdr(os.path.join(tmp, "old_hall_4k.hdr"), 4096, 2048)
if "bust" in scene:
    api.write_bust_obj(os.path.join(tmp, "marble_bust_01.obj"), 1.0)
hs = api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", scene), tmp)
dev = api.DeviceScene(hs.flat_ptr, 0)
cam = hs.camera(W, H)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
p = api.default_params(W, H, spp, quirks=quirks, seed=0, timing=True)
ref = None
for v in values:
    if v == "default":
        os.environ.pop(knob, None)
    else:
        os.environ[knob] = v
    dev.render_stripes_device(cam, p, H, 0, 1, out.data_ptr(), stream)   # warm-up
    torch.cuda.synchronize()
    dev.stats()
    for _ in range(3):
        dev.render_stripes_device(cam, p, H, 0, 1, out.data_ptr(), stream)
    torch.cuda.synchronize()
    st = dev.stats()
    img = out.clone()
    if ref is None:
        ref = img
    same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
    print(f"{knob}={v}: frame {st.kernel_ms / st.launches:.2f} ms  traversal {st.traversal_ms / st.launches:.2f} ms  image==first {same}", flush=True)
