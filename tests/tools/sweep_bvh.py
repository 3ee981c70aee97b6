"""Experiment tool (GPU box): frame / traversal time of one scene for several BVH builder settings
(HRT_BVH_MAX_LEAF, HRT_BVH_TRI_COST are read by the host builder when the scene is loaded).
usage: python3 tests/tools/sweep_bvh.py [scene.yaml] [W H spp quirks]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hobbyraytracer_amd import api  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "teapot_scene.yaml"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (640, 640, 100)
quirks = api.QUIRKS_FIXED if (len(sys.argv) > 5 and sys.argv[5] == "fixed") else api.QUIRKS_REFERENCE
tmp = tempfile.mkdtemp(prefix="hrt_sweep_")
api.write_teapot_obj(os.path.join(tmp, "teapot.obj"), 1.0)
api.write_hall_hdr(os.path.join(tmp, "old_hall_4k.hdr"), 4096, 2048)
if "bust" in scene:
    api.write_bust_obj(os.path.join(tmp, "marble_bust_01.obj"), 1.0)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ref = None
combos = [tuple(float(x) for x in c.split(":")) for c in os.environ.get("SWEEP", "4:1.3,1:1.3,2:1.3,3:1.3,4:1.0,4:2.0,8:1.3").split(",")]
for max_leaf, tri_cost in combos:
    max_leaf = int(max_leaf)
    os.environ["HRT_BVH_MAX_LEAF"] = str(max_leaf)
    os.environ["HRT_BVH_TRI_COST"] = str(tri_cost)
    hs = api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", scene), tmp)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    cam = hs.camera(W, H)
    ps = api.default_params(W, H, spp, quirks=quirks, seed=0, stats=True)
    p = api.default_params(W, H, spp, quirks=quirks, seed=0, timing=True)
    dev.render_stripes_device(cam, ps, H, 0, 1, out.data_ptr(), stream)
    torch.cuda.synchronize()
    sc = dev.stats()
    img = out.clone()
    if ref is None:
        ref = img
    same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
    for _ in range(3):
        dev.render_stripes_device(cam, p, H, 0, 1, out.data_ptr(), stream)
    torch.cuda.synchronize()
    st = dev.stats()
    print(f"max_leaf {max_leaf} tri_cost {tri_cost}: frame {st.kernel_ms / st.launches:.2f} ms  traversal {st.traversal_ms / st.launches:.2f} ms  "
          f"box/ray {sc.box_tests / sc.rays:.2f} tri/ray {sc.tri_tests / sc.rays:.2f} depth {hs.bvh_depth(0)} image==first {same}", flush=True)
    del dev, hs
