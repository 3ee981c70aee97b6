"""usage (GPU box, repo root): python3 tests/tools/fuzz_lbvh.py N [seed] [lbvh|sah] -- random triangle soups through hrt_bvh_build_device / hrt_bvh_build_sah
(csrc/hrt_lbvh.hip): sizes 3..20000, coordinates from 1e-6 to 1e30 wide, clustered / coincident / degenerate triangles, signed zeros.
Every tree is walked by tests/test_gpu_scenes.py _lbvh_check (permutation, leaf sizes, child boxes bit for bit, depth); trees deeper than
the traversal stack are counted, not failed."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hobbyraytracer_amd import api
from tests.test_gpu_scenes import _lbvh_check
N = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
algo = sys.argv[3] if len(sys.argv) > 3 else "lbvh"
r = np.random.default_rng(seed)
bad = deep = 0
for it in range(N):
    n = int(r.choice([3, 4, 5, 7, 17, 100, 1000, int(r.integers(3, 20000))]))
    kind = int(r.integers(0, 6))
    scale = float(10.0 ** r.uniform(-6, 30)) if r.random() < 0.5 else 1.0
    c = r.uniform(-1, 1, (n, 1, 3))
    if kind == 1: c[:] = c[0]                                   # every centroid in one place (up to the jitter)
    if kind == 2: c = np.round(c * 4) / 4                        # a lattice: many equal Morton codes
    if kind == 3: c[:, :, r.integers(0, 3)] = 0.0                # flat
    tri = c + r.normal(scale=float(r.choice([0.0, 1e-7, 0.01, 0.3])), size=(n, 3, 3))
    if kind == 4: tri[::3, 1] = tri[::3, 0]                      # zero-area triangles
    if kind == 5: tri = np.repeat(tri[: max(1, n // 10)], 10, 0)[:n]   # every triangle ten times
    tri = (tri * scale).astype(np.float32).reshape(-1, 9)
    if r.random() < 0.1: tri[r.integers(0, len(tri)), r.integers(0, 9)] = -0.0
    if not np.isfinite(tri).all() or np.abs(tri).max() > 8e37: continue
    ml = int(r.choice([1, 2, 2, 4, 8]))
    if len(tri) <= ml: continue
    try:
        nodes, order, depth = api.bvh_build_device(tri, ml, algo=algo)
        _lbvh_check(tri, nodes, order, depth, ml)
        deep += depth > 31
    except api.HrtError as e:
        if e.status == api.HRT_ERR_UNSUPPORTED and algo == "sah":      # a large node needs the host's median split: handed back, by design
            handed_back = globals().get("handed_back", 0) + 1
            continue
        bad += 1
        print("BAD", it, n, kind, scale, ml, repr(e)[:200], flush=True)
    except Exception as e:
        bad += 1
        print("BAD", it, n, kind, scale, ml, repr(e)[:200], flush=True)
    if (it + 1) % 200 == 0: print("progress", it + 1, "bad", bad, "deeper than 31:", deep, flush=True)
print("done", N, "bad", bad, "deeper than 31:", deep, "handed back to the host:", globals().get("handed_back", 0))
