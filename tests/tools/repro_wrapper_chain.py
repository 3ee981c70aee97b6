"""Reproducer harness for the wrapper-chain miscompile (GPU box, repo root):
    python3 tests/tools/repro_wrapper_chain.py [path/to/libhrt_hip_variant.so]
Renders tests/scene_helpers.py wrapper_chain_scene for chains of every length on each render path and prints how many film
pixels differ from the oracle.  With a library built with -DHRT_XF_FLAT (world_rec's chain walk as four independent predicated
blocks, the shape that came out wrong) the chains of exactly three wrappers differ on the wavefront paths; the shipped shape
(nested ifs) is clean everywhere."""
import os, sys, tempfile, pathlib, shutil
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    from hobbyraytracer_amd import __file__ as _pkg
    lib = os.path.join(os.path.dirname(_pkg), "lib", "libhrt_hip.so")
    backup = lib + ".orig"
    shutil.copy(lib, backup)
    shutil.copy(sys.argv[1], lib + ".new")
    os.replace(lib + ".new", lib)          # (a new inode: nothing that has the old file mapped is disturbed)
import numpy as np
try:
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import wrapper_chain_scene
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)
    for chain in ["YQ", "YQS", "YQT", "QST", "YST", "YQST"]:
        d = pathlib.Path(tempfile.mkdtemp())
        os.dup2(devnull, 1)
        try:
            hs = api.HostScene(wrapper_chain_scene(d, chain), str(d))
            dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
            cam = hs.camera(48, 48)
            out = []
            for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
                ref, _ = world.render_tile(cam, api.default_params(48, 48, 4, quirks=q))
                for name, tail, mega in (("pipeline", "1000", False), ("tail", "1", False), ("megakernel", "1", True)):
                    os.environ["HRT_WF_TAIL_ROUND"] = tail
                    img, _ = dev.render_tile(cam, api.default_params(48, 48, 4, quirks=q, megakernel=mega))
                    out.append((q, name, int((img.view(np.uint32) != ref.view(np.uint32)).any(2).sum())))
            dev.close()
        finally:
            os.dup2(saved, 1)
        print(chain, " ".join(f"q{q}:{n}={b}" for q, n, b in out), flush=True)
finally:
    if len(sys.argv) > 1:
        os.replace(backup, lib)
