"""usage (GPU box, repo root): python3 tests/tools/bvh_builders.py -- the host's binned-SAH builder against hrt_bvh_build_device / hrt_bvh_build_sah (LBVH / the same binned SAH on
the GPU, csrc/hrt_lbvh.hip, hrt_sahbvh.hip): scene load time (parse + import + build + flatten), the device build alone, tree size and depth, and what the
tree costs a render (headline teapot frame, bust scene at 1024x1024x16; wavefront pipeline, reference quirks)."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api
d = tempfile.mkdtemp()
api.write_teapot_obj(d + "/teapot.obj", 1.0); api.write_bust_obj(d + "/marble_bust_01.obj", 1.0); api.write_hall_hdr(d + "/old_hall_4k.hdr", 512, 256)
for scene, W, H, spp in (("teapot_scene.yaml", 640, 640, 100), ("bust_scene.yaml", 1024, 1024, 16)):
    for builder in ("sah", "lbvh", "gpu-sah"):
        api.use_device_bvh_builder(builder != "sah", algo={"sah": "lbvh", "lbvh": "lbvh", "gpu-sah": "sah"}[builder])
        try:
            api.HostScene("tests/golden/scenes/" + scene, d)        # warm (file cache, device context)
            t0 = time.time(); hs = api.HostScene("tests/golden/scenes/" + scene, d); t_load = time.time() - t0
        finally:
            api.use_device_bvh_builder(False)
        pos = np.asarray(hs.mesh_arrays(0)[0], dtype=np.float32).reshape(-1, 9)
        algo = {"sah": "lbvh", "lbvh": "lbvh", "gpu-sah": "sah"}[builder]
        api.bvh_build_device(pos, 2, algo=algo)                     # (the first call of a process loads the code object)
        t0 = time.time(); api.bvh_build_device(pos, 2, algo=algo); t_dev = time.time() - t0
        dev = api.DeviceScene(hs.flat_ptr, 0)
        cam = hs.camera(W, H)
        p = api.default_params(W, H, spp, stats=True)
        dev.render_tile(cam, p)
        img, st = dev.render_tile(cam, p)
        pt = api.default_params(W, H, spp, timing=True)
        dev.render_tile(cam, pt); dev.stats()
        t0 = time.time(); dev.render_tile(cam, pt); t_frame = time.time() - t0
        dev.stats()
        print("%-18s %-7s tris %7d nodes %7d depth %2d | load %7.1f ms (device build alone %6.1f ms) | box tests / segment %6.2f tri tests %5.2f | frame %7.1f ms" % (
            scene, builder, pos.shape[0], hs.flat.n_nodes, hs.bvh_depth(0), 1e3 * t_load, 1e3 * t_dev, st.box_tests / st.rays, st.tri_tests / st.rays, 1e3 * t_frame), flush=True)
        dev.close()
