"""usage: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tests/tools/fuzz_flat.py N [seed] [dir with an
ASan build of flatcpu_py.py + libflatcpu.so] -- a fuzz campaign, not a test, for the claim "what hrt_scene_create accepts, the
device code reads in bounds".  Flat scenes (include/hrt.h) of the golden YAML files get random fields overwritten (indices,
counts, kinds, offsets, floats -> -1 / 0 / huge / NaN / inf); hrt_scene_create validates before it touches a device, so on a box
without a GPU it answers HRT_ERR_NO_DEVICE exactly when the scene passed.  Every scene that passes is rendered and probed by
the device header compiled for the host with AddressSanitizer (tests/tools/flat_on_cpu.cpp): an out-of-bounds read there is
an out-of-bounds read on the GPU."""
import ctypes as C, os, signal, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3: sys.path.insert(0, sys.argv[3])
else: sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
from hobbyraytracer_amd import api
import flatcpu_py

N = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.random.default_rng(seed)
d = tempfile.mkdtemp()
api.write_teapot_obj(os.path.join(d, "teapot.obj"), 0.05)
api.write_hall_hdr(os.path.join(d, "old_hall_4k.hdr"), 32, 16)
api.write_bust_obj(os.path.join(d, "marble_bust_01.obj"), 0.02)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
hosts = []
for name in ("teapot_scene", "three_meshes", "material_zoo", "cornell_box", "triangles", "bust_scene"):
    try: hosts.append((name, api.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", name + ".yaml"), d)))
    except Exception as e: pass
os.dup2(saved, 1)
print("scenes:", [n for n, _ in hosts], flush=True)

INTS = [-1, 0, 1, 2, 3, 7, 8, 255, 65535, 65536, 2**31 - 1, -2**31, 12345]
FLOATS = [float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e38, -1e38, 1e-38, 1.0, -1.0, 3e38]

def clone(flat):
    """Deep copy of a flat scene into ctypes arrays this script owns."""
    f = api.FlatScene(); C.memmove(C.byref(f), C.byref(flat), C.sizeof(api.FlatScene)); keep = []
    def arr(ptr, typ, n):
        a = (typ * max(int(n), 1))()
        if n and ptr: C.memmove(a, ptr, C.sizeof(typ) * int(n))
        keep.append(a); return a
    f.prims = C.cast(arr(flat.prims, api.Prim, flat.n_prims), C.POINTER(api.Prim))
    f.materials = C.cast(arr(flat.materials, api.Material, flat.n_materials), C.POINTER(api.Material))
    f.textures = C.cast(arr(flat.textures, api.Texture, flat.n_textures), C.POINTER(api.Texture))
    f.meshes = C.cast(arr(flat.meshes, api.Mesh, flat.n_meshes), C.POINTER(api.Mesh))
    f.nodes = C.cast(arr(flat.nodes, api.BvhNode, flat.n_nodes), C.POINTER(api.BvhNode))
    f.tri_pos = C.cast(arr(flat.tri_pos, C.c_float, flat.n_tris * 9), C.POINTER(C.c_float))
    f.tri_nrm = C.cast(arr(flat.tri_nrm, C.c_float, flat.n_tris * 9), C.POINTER(C.c_float))
    f.tri_uv = C.cast(arr(flat.tri_uv, C.c_float, flat.n_tris * 6), C.POINTER(C.c_float))
    if flat.tri_box: f.tri_box = C.cast(arr(flat.tri_box, C.c_float, flat.n_tris * 6), C.POINTER(C.c_float))
    if flat.tri_ref_order: f.tri_ref_order = C.cast(arr(flat.tri_ref_order, C.c_uint32, flat.n_tris), C.POINTER(C.c_uint32))
    f.texels_u8 = C.cast(arr(flat.texels_u8, C.c_uint8, flat.n_texels_u8), C.POINTER(C.c_uint8))
    f.texels_f32 = C.cast(arr(flat.texels_f32, C.c_float, flat.n_texels_f32), C.POINTER(C.c_float))
    return f, keep

def mutate_struct(s):
    """Overwrite one random leaf field of a ctypes structure (recursing into nested structs / arrays)."""
    name, typ = s._fields_[int(r.integers(0, len(s._fields_)))]
    v = getattr(s, name)
    if isinstance(v, C.Structure): return mutate_struct(v)
    if isinstance(v, C.Array):
        i = int(r.integers(0, len(v)))
        if isinstance(v[i], C.Structure): return mutate_struct(v[i])
        v[i] = FLOATS[int(r.integers(0, len(FLOATS)))] if isinstance(v[i], float) else INTS[int(r.integers(0, len(INTS)))]
        return
    if isinstance(v, float): setattr(s, name, FLOATS[int(r.integers(0, len(FLOATS)))])
    elif isinstance(v, int):
        x = INTS[int(r.integers(0, len(INTS)))]
        try: setattr(s, name, x)
        except Exception: pass
        if typ in (C.c_uint32, C.c_uint64) and x < 0: setattr(s, name, x & 0xFFFFFFFF)

def mutate(f):
    what = int(r.integers(0, 11))
    if what == 0 and f.n_prims: mutate_struct(f.prims[int(r.integers(0, f.n_prims))])
    elif what == 1 and f.n_materials: mutate_struct(f.materials[int(r.integers(0, f.n_materials))])
    elif what == 2 and f.n_textures: mutate_struct(f.textures[int(r.integers(0, f.n_textures))])
    elif what == 3 and f.n_meshes: mutate_struct(f.meshes[int(r.integers(0, f.n_meshes))])
    elif what == 4 and f.n_nodes: mutate_struct(f.nodes[int(r.integers(0, f.n_nodes))])
    elif what == 5 and f.n_tris: f.tri_pos[int(r.integers(0, f.n_tris * 9))] = FLOATS[int(r.integers(0, len(FLOATS)))]
    elif what == 6 and f.n_tris: f.tri_nrm[int(r.integers(0, f.n_tris * 9))] = FLOATS[int(r.integers(0, len(FLOATS)))]
    elif what == 7 and f.n_tris and f.tri_box: f.tri_box[int(r.integers(0, f.n_tris * 6))] = FLOATS[int(r.integers(0, len(FLOATS)))]
    elif what == 8 and f.n_tris and f.tri_ref_order: f.tri_ref_order[int(r.integers(0, f.n_tris))] = int(r.integers(0, 2**32))
    elif what == 9:   # the scene-level counts and the background index (counts only DOWN: the arrays are as long as they are)
        k = int(r.integers(0, 8))
        if k == 0: f.background_tex = INTS[int(r.integers(0, len(INTS)))]
        elif k == 1 and f.n_prims: f.n_prims = int(r.integers(0, f.n_prims + 1))
        elif k == 2 and f.n_materials: f.n_materials = int(r.integers(0, f.n_materials + 1))
        elif k == 3 and f.n_textures: f.n_textures = int(r.integers(0, f.n_textures + 1))
        elif k == 4 and f.n_meshes: f.n_meshes = int(r.integers(0, f.n_meshes + 1))
        elif k == 5 and f.n_tris: f.n_tris = int(r.integers(0, f.n_tris + 1))
        elif k == 6 and f.n_nodes: f.n_nodes = int(r.integers(0, f.n_nodes + 1))
        elif k == 7: f.n_texels_u8 = int(r.integers(0, f.n_texels_u8 + 1)); f.n_texels_f32 = int(r.integers(0, f.n_texels_f32 + 1))
    elif what == 10 and f.n_tris: f.tri_uv[int(r.integers(0, f.n_tris * 6))] = FLOATS[int(r.integers(0, len(FLOATS)))]

class Timeout(Exception): pass
def on_alarm(sig, frm): raise Timeout()
signal.signal(signal.SIGALRM, on_alarm)

accepted = refused = 0
reasons = {}
for it in range(N):
    name, hs = hosts[int(r.integers(0, len(hosts)))]
    f, keep = clone(hs.flat)
    for _ in range(int(r.integers(1, 4))): mutate(f)
    h = C.c_void_p()
    st = api._hip.hrt_scene_create(C.byref(f), 0, C.byref(h))
    if st != api.HRT_ERR_NO_DEVICE:
        if st == api.HRT_OK: api._hip.hrt_scene_destroy(h)   # (a GPU box: validation passed too)
        else:
            refused += 1; msg = api._hip.hrt_last_error().decode()[:40]; reasons[msg] = reasons.get(msg, 0) + 1
            continue
    accepted += 1
    signal.alarm(60)
    try:
        flat = flatcpu_py.FlatCpu(C.pointer(f))
        cam = hs.camera(12, 12)
        for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
            flat.render_tile(cam, api.default_params(12, 12, 2, quirks=q, stats=True))
        o = r.uniform(-4, 4, (300, 3)).astype(np.float32); dd = r.normal(size=(300, 3)).astype(np.float32)
        flat.closest_hit(api.default_params(8, 8, 1), o, dd)
        flat.close()
    except Timeout:
        print("TIMEOUT (an endless walk?) on mutant", it, "of", name, flush=True)
    finally:
        signal.alarm(0)
    if it % 500 == 499: print("progress", it + 1, "accepted", accepted, "refused", refused, flush=True)
print("done", N, "accepted", accepted, "refused", refused)
for k, v in sorted(reasons.items(), key=lambda kv: -kv[1])[:25]: print("  %6d  %s" % (v, k))
