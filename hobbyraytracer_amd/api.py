"""ctypes bindings of include/hrt.h (libhrt_hip.so) and include/hrt_host.h (libhrt_host.so).

Struct layouts mirror the headers field for field; ``tests/test_abi.py`` checks the sizes
against ``sizeof`` values compiled from the headers.  Every wrapper raises :class:`HrtError`
on a non-zero ``hrt_status`` — nothing here falls back to a CPU path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_LIB_PATH = os.environ.get("HRT_HIP_LIB") or os.path.join(LIB_DIR, "libhrt_hip.so")   # HRT_HIP_LIB: an instrumented build of the
# same library (-DHRT_DEBUG_BOUNDS, profiling variants; tests/tools), never another implementation
HOST_LIB_PATH = os.path.join(LIB_DIR, "libhrt_host.so")
CLI_PATH = os.path.join(_HERE, "bin", "hobbyraytracer")


class HrtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"hrt status {status}: {message}")
        self.status = status


# ---------------------------------------------------------------- constants (hrt.h)
HRT_OK, HRT_ERR_INVALID, HRT_ERR_HIP, HRT_ERR_NO_DEVICE, HRT_ERR_OOM, HRT_ERR_IO, HRT_ERR_PARSE, HRT_ERR_UNSUPPORTED = range(8)
PRIM_SPHERE, PRIM_XY_RECT, PRIM_XZ_RECT, PRIM_YZ_RECT, PRIM_BOX, PRIM_MESH, PRIM_MEDIUM, PRIM_TRIANGLE = range(8)
XF_TRANSLATE, XF_SCALE, XF_ROTATE_QUAT, XF_ROTATE_Y = range(4)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC, MAT_PBR, MAT_UVTEST = range(7)
TEX_SOLID, TEX_CHECKER, TEX_IMAGE, TEX_ENV = range(4)
MAX_XFORMS = 4
Q1_ROTQ_NORMALIZE, Q2_TRI_NO_TMIN, Q3_TRI_NO_FACE, Q4_SHEAR_FROM_ORIGIN = 1, 2, 4, 8
QUIRKS_REFERENCE, QUIRKS_FIXED = 0xF, 0x0
FLAG_STATS, FLAG_MEGAKERNEL, FLAG_TIMING, FLAG_THIN_LENS, FLAG_PROGRESS = 1, 2, 4, 8, 16


# ---------------------------------------------------------------- structs (hrt.h)
class Xform(C.Structure):
    _fields_ = [("kind", C.c_int32), ("v", C.c_float * 4)]


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("mesh", C.c_int32), ("boundary_kind", C.c_int32),
                ("p", C.c_float * 9), ("density", C.c_float), ("n_xforms", C.c_int32), ("xf", Xform * MAX_XFORMS)]


class MatVec3(C.Structure):
    _fields_ = [("tex", C.c_int32), ("c", C.c_float * 3)]


class MatScalar(C.Structure):
    _fields_ = [("tex", C.c_int32), ("c", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", MatVec3), ("s0", MatScalar), ("s1", MatScalar), ("mix_tex", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("c", C.c_float * 3), ("even", C.c_int32), ("odd", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("channels", C.c_int32), ("_pad", C.c_int32), ("offset", C.c_uint64)]


class Mesh(C.Structure):
    _fields_ = [("tri_first", C.c_uint32), ("tri_count", C.c_uint32), ("node_first", C.c_uint32), ("node_count", C.c_uint32)]


class BvhNode(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("c0_min_x", "c0_max_x", "c0_min_y", "c0_max_y", "c1_min_x", "c1_max_x", "c1_min_y",
                                         "c1_max_y", "c0_min_z", "c0_max_z", "c1_min_z", "c1_max_z")] + \
               [("child0", C.c_int32), ("child1", C.c_int32), ("_pad0", C.c_int32), ("_pad1", C.c_int32)]


class FlatScene(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("prims", C.POINTER(Prim)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("n_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh)),
                ("n_tris", C.c_uint64), ("tri_pos", C.POINTER(C.c_float)), ("tri_nrm", C.POINTER(C.c_float)),
                ("tri_uv", C.POINTER(C.c_float)), ("tri_box", C.POINTER(C.c_float)), ("tri_ref_order", C.POINTER(C.c_uint32)),
                ("n_nodes", C.c_uint64), ("nodes", C.POINTER(BvhNode)),
                ("n_texels_u8", C.c_uint64), ("texels_u8", C.POINTER(C.c_uint8)),
                ("n_texels_f32", C.c_uint64), ("texels_f32", C.POINTER(C.c_float)),
                ("background_tex", C.c_int32), ("_pad", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("lens_u", C.c_float * 3), ("lens_v", C.c_float * 3), ("lens_radius", C.c_float)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("max_depth", C.c_int32),
                ("t_min", C.c_float), ("quirks", C.c_uint32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("flags", C.c_uint32)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("mesh_hits", C.c_uint64), ("env_lookups", C.c_uint64), ("kernel_ms", C.c_double), ("launches", C.c_uint64),
                ("traversal_ms", C.c_double), ("traversal_launches", C.c_uint64),
                ("traversal_box_tests", C.c_uint64), ("traversal_tri_tests", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    def algorithmic_bytes(self, n_pixels=0):
        """32 B per box tested, 36 B per triangle tested, 60 B of attributes per
        mesh hit, 12 B per fp32 environment lookup, 12 B per pixel written."""
        return 32 * self.box_tests + 36 * self.tri_tests + 60 * self.mesh_hits + 12 * self.env_lookups + 12 * n_pixels


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_int32), ("tri", C.c_int32), ("front_face", C.c_int32), ("p", C.c_float * 3),
                ("normal", C.c_float * 3), ("u", C.c_float), ("v", C.c_float)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<i4"), ("tri", "<i4"), ("front_face", "<i4"), ("p", "<f4", 3),
                      ("normal", "<f4", 3), ("u", "<f4"), ("v", "<f4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit)

HIP_SYMBOLS = ["hrt_device_count", "hrt_scene_create", "hrt_scene_destroy", "hrt_render_tile", "hrt_render_stripes_device",
               "hrt_render_stripes", "hrt_render_stripes_accumulate_device", "hrt_render_stripes_accumulate", "hrt_stripe_rows", "hrt_stripe_row_index", "hrt_scene_stats", "hrt_resolve_u8",
               "hrt_resolve_u8_device", "hrt_closest_hit", "hrt_math_probe", "hrt_status_str", "hrt_last_error", "hrt_version",
               "hrt_multi_create", "hrt_multi_destroy", "hrt_multi_devices", "hrt_multi_uses_rccl", "hrt_multi_render", "hrt_bvh_build_device", "hrt_bvh_build_sah",
               "hrt_debug_bounds_violations", "hrt_scene_progress", "hrt_multi_progress"]
HOST_SYMBOLS = ["hrt_host_load_yaml", "hrt_host_free", "hrt_host_flat", "hrt_host_film", "hrt_host_camera", "hrt_host_bvh_depth",
                "hrt_default_params", "hrt_asset_write_teapot_obj", "hrt_asset_write_bust_obj", "hrt_asset_write_hall_hdr",
                "hrt_host_write_image", "hrt_host_read_hdr", "hrt_host_read_png", "hrt_host_read_jpeg", "hrt_host_write_hdr", "hrt_host_write_pfm", "hrt_host_read_pfm", "hrt_host_last_error", "hrt_host_set_bvh_builder"]


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' or `make`). "
                          "There is no fallback implementation.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host = _load(HOST_LIB_PATH)
_hip = _load(HIP_LIB_PATH)

_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p

_hip.hrt_status_str.restype = C.c_char_p
_hip.hrt_last_error.restype = C.c_char_p
_hip.hrt_version.restype = C.c_char_p
_hip.hrt_device_count.argtypes = [C.POINTER(C.c_int)]
_hip.hrt_scene_create.argtypes = [C.POINTER(FlatScene), C.c_int, C.POINTER(_vp)]
_hip.hrt_scene_destroy.argtypes = [_vp]
_hip.hrt_scene_destroy.restype = None
_hip.hrt_render_tile.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), Rect, _fp, C.POINTER(Stats)]
_hip.hrt_render_stripes_device.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, _vp, _vp]
_hip.hrt_render_stripes.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, _fp, C.POINTER(Stats)]
_hip.hrt_render_stripes_accumulate_device.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, _vp,
                                                      C.c_int32, C.c_int32, _vp]
_hip.hrt_render_stripes_accumulate.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, _fp,
                                               C.c_int32, C.c_int32, C.POINTER(Stats)]
_hip.hrt_stripe_rows.argtypes = [C.c_int32] * 4
_hip.hrt_stripe_rows.restype = C.c_int32
_hip.hrt_stripe_row_index.argtypes = [C.c_int32] * 5
_hip.hrt_stripe_row_index.restype = C.c_int32
_hip.hrt_scene_stats.argtypes = [_vp, C.POINTER(Stats)]
_hip.hrt_resolve_u8.argtypes = [_vp, _fp, C.c_int64, _u8p]
_hip.hrt_resolve_u8_device.argtypes = [_vp, _vp, C.c_int64, _vp, _vp]
_hip.hrt_closest_hit.argtypes = [_vp, C.POINTER(Params), C.c_int64, _fp, _fp, C.c_float, C.c_float, C.c_uint32, C.POINTER(Hit)]
_hip.hrt_multi_create.argtypes = [C.POINTER(FlatScene), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_vp)]
_hip.hrt_multi_destroy.argtypes = [_vp]
_hip.hrt_multi_destroy.restype = None
_hip.hrt_multi_devices.argtypes = [_vp]
_hip.hrt_multi_devices.restype = C.c_int32
_hip.hrt_multi_uses_rccl.argtypes = [_vp]
_hip.hrt_multi_uses_rccl.restype = C.c_int32
_hip.hrt_multi_render.argtypes = [_vp, C.POINTER(Camera), C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _u8p, C.POINTER(Stats)]
_hip.hrt_math_probe.argtypes = [C.c_int, C.c_int32, C.c_int64, _fp, _fp, _fp]

_host.hrt_host_last_error.restype = C.c_char_p
_host.hrt_host_load_yaml.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_vp)]
_host.hrt_host_free.argtypes = [_vp]
_host.hrt_host_free.restype = None
_host.hrt_host_flat.argtypes = [_vp]
_host.hrt_host_flat.restype = C.POINTER(FlatScene)
_host.hrt_host_film.argtypes = [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
_host.hrt_host_camera.argtypes = [_vp, C.c_int32, C.c_int32, C.POINTER(Camera)]
_host.hrt_host_bvh_depth.argtypes = [_vp, C.c_int32]
_host.hrt_host_bvh_depth.restype = C.c_int32
_host.hrt_default_params.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32]
_host.hrt_default_params.restype = None
_host.hrt_asset_write_teapot_obj.argtypes = [C.c_char_p, C.c_double]
_host.hrt_asset_write_teapot_obj.restype = C.c_int64
_host.hrt_asset_write_bust_obj.argtypes = [C.c_char_p, C.c_double]
_host.hrt_asset_write_bust_obj.restype = C.c_int64
_host.hrt_asset_write_hall_hdr.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
_host.hrt_host_write_image.argtypes = [C.c_char_p, _u8p, C.c_int32, C.c_int32]
_host.hrt_host_read_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp, C.c_int64]
_host.hrt_host_read_png.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u8p, C.c_int64]
_host.hrt_host_read_jpeg.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u8p, C.c_int64]
_host.hrt_host_write_hdr.argtypes = [C.c_char_p, _fp, C.c_int32, C.c_int32]
_host.hrt_host_write_pfm.argtypes = [C.c_char_p, _fp, C.c_int32, C.c_int32]
_host.hrt_host_read_pfm.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _fp, C.c_int64]


def _check(st):
    if st != HRT_OK:
        raise HrtError(st, f"{_hip.hrt_status_str(st).decode()}: {_hip.hrt_last_error().decode()}")


def _check_host(st):
    if st != HRT_OK:
        raise HrtError(st, f"{_hip.hrt_status_str(st).decode()}: {_host.hrt_host_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=_fp):
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------- host side
def default_params(width, height, samples, quirks=QUIRKS_REFERENCE, seed=0, max_depth=50, stats=False, megakernel=False, timing=False,
                   thin_lens=False, progress=False):
    p = Params()
    _host.hrt_default_params(C.byref(p), width, height, samples)
    p.quirks = quirks
    p.seed_lo = seed & 0xFFFFFFFF
    p.seed_hi = (seed >> 32) & 0xFFFFFFFF
    p.max_depth = max_depth
    p.flags = (FLAG_STATS if stats else 0) | (FLAG_MEGAKERNEL if megakernel else 0) | (FLAG_TIMING if timing else 0) | \
              (FLAG_THIN_LENS if thin_lens else 0) | (FLAG_PROGRESS if progress else 0)
    return p


class HostScene:
    """Scene::loadScene + flatten (scene.cpp:127-379)."""

    def __init__(self, yaml_path, asset_dir=None):
        self._h = None
        h = _vp()
        _check_host(_host.hrt_host_load_yaml(os.fsencode(yaml_path), os.fsencode(asset_dir) if asset_dir else None, C.byref(h)))
        self._h = h
        self.flat_ptr = _host.hrt_host_flat(h)
        self.flat = self.flat_ptr.contents

    def close(self):
        if self._h:
            _host.hrt_host_free(self._h)
            self._h = None

    def __del__(self):
        try:                     # at interpreter shutdown the module globals may already be gone
            self.close()
        except Exception:
            pass

    @property
    def film(self):
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(1024)
        _check_host(_host.hrt_host_film(self._h, C.byref(w), C.byref(h), C.byref(s), buf, 1024))
        return w.value, h.value, s.value, buf.value.decode()

    def camera(self, width=None, height=None):
        fw, fh, _, _ = self.film
        cam = Camera()
        _check_host(_host.hrt_host_camera(self._h, width or fw, height or fh, C.byref(cam)))
        return cam

    def bvh_depth(self, mesh=0):
        return _host.hrt_host_bvh_depth(self._h, mesh)

    def mesh_arrays(self, mesh=0):
        m = self.flat.meshes[mesh]
        n = m.tri_count
        pos = np.ctypeslib.as_array(self.flat.tri_pos, shape=(self.flat.n_tris * 9,))[m.tri_first * 9:(m.tri_first + n) * 9].reshape(n, 3, 3)
        nrm = np.ctypeslib.as_array(self.flat.tri_nrm, shape=(self.flat.n_tris * 9,))[m.tri_first * 9:(m.tri_first + n) * 9].reshape(n, 3, 3)
        uv = np.ctypeslib.as_array(self.flat.tri_uv, shape=(self.flat.n_tris * 6,))[m.tri_first * 6:(m.tri_first + n) * 6].reshape(n, 3, 2)
        return pos, nrm, uv


def write_teapot_obj(path, detail=1.0):
    n = _host.hrt_asset_write_teapot_obj(os.fsencode(path), detail)
    if n < 0:
        raise HrtError(HRT_ERR_IO, f"cannot write {path}")
    return n


def write_bust_obj(path, detail=1.0):
    n = _host.hrt_asset_write_bust_obj(os.fsencode(path), detail)
    if n < 0:
        raise HrtError(HRT_ERR_IO, f"cannot write {path}")
    return n


def write_hall_hdr(path, width=4096, height=2048):
    _check_host(_host.hrt_asset_write_hall_hdr(os.fsencode(path), width, height))


def write_image(path, rgb8):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = a.shape
    _check_host(_host.hrt_host_write_image(os.fsencode(path), _ptr(a, _u8p), w, h))


def read_hdr(path):
    w, h = C.c_int32(), C.c_int32()
    _check_host(_host.hrt_host_read_hdr(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.float32)
    _check_host(_host.hrt_host_read_hdr(os.fsencode(path), C.byref(w), C.byref(h), _ptr(out), out.size))
    return out


def write_hdr(path, rgb):
    a = _f32(rgb)
    h, w, _ = a.shape
    _check_host(_host.hrt_host_write_hdr(os.fsencode(path), _ptr(a), w, h))


def read_jpeg(path):
    w, h = C.c_int32(), C.c_int32()
    _check_host(_host.hrt_host_read_jpeg(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    _check_host(_host.hrt_host_read_jpeg(os.fsencode(path), C.byref(w), C.byref(h), _ptr(out, _u8p), out.size))
    return out


def write_pfm(path, rgb):
    a = _f32(rgb)
    h, w, _ = a.shape
    _check_host(_host.hrt_host_write_pfm(os.fsencode(path), _ptr(a), w, h))


def read_pfm(path):
    w, h = C.c_int32(), C.c_int32()
    _check_host(_host.hrt_host_read_pfm(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.float32)
    _check_host(_host.hrt_host_read_pfm(os.fsencode(path), C.byref(w), C.byref(h), _ptr(out), out.size))
    return out


def read_png(path):
    w, h = C.c_int32(), C.c_int32()
    _check_host(_host.hrt_host_read_png(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    _check_host(_host.hrt_host_read_png(os.fsencode(path), C.byref(w), C.byref(h), _ptr(out, _u8p), out.size))
    return out


# ---------------------------------------------------------------- device side
def device_count():
    n = C.c_int()
    st = _hip.hrt_device_count(C.byref(n))
    return n.value if st == HRT_OK else 0


def use_device_bvh_builder(enable=True, device=0, algo="lbvh"):
    """hrt_host_set_bvh_builder: scenes loaded from now on get their meshes' culling trees from the GPU -- hrt_bvh_build_device
    (algo "lbvh": a Morton-ordered LBVH, fastest) or hrt_bvh_build_sah ("sah": the host builder's own binned-SAH tree, built on
    the device) -- instead of the host's builder."""
    _host.hrt_host_set_bvh_builder.argtypes = [C.c_void_p, C.c_int]
    _host.hrt_host_set_bvh_builder.restype = None
    fn = {"lbvh": _hip.hrt_bvh_build_device, "sah": _hip.hrt_bvh_build_sah}[algo]
    _host.hrt_host_set_bvh_builder(C.cast(fn, C.c_void_p) if enable else None, device)


def bvh_build_device(tri_pos, max_leaf=2, device=0, algo="lbvh"):
    """hrt_bvh_build_device on an (n, 9) float32 array: (nodes as an (n_nodes, 16) uint32 view of hrt_bvh_node, order, depth)."""
    pos = np.ascontiguousarray(tri_pos, dtype=np.float32).reshape(-1, 9)
    n = pos.shape[0]
    nodes = np.zeros((max(n - 1, 1), 16), dtype=np.uint32)
    order = np.zeros(n, dtype=np.uint32)
    n_nodes, depth = C.c_uint32(), C.c_int32()
    f = {"lbvh": _hip.hrt_bvh_build_device, "sah": _hip.hrt_bvh_build_sah}[algo]
    f.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    _check(f(device, pos.ctypes.data, n, max_leaf, nodes.ctypes.data, C.addressof(n_nodes), order.ctypes.data, C.addressof(depth)))
    return nodes[:n_nodes.value], order, depth.value


def stripe_rows(height, rows_per_block, rank, n_ranks):
    return _hip.hrt_stripe_rows(height, rows_per_block, rank, n_ranks)


def stripe_row_index(height, rows_per_block, rank, n_ranks, local_row):
    return _hip.hrt_stripe_row_index(height, rows_per_block, rank, n_ranks, local_row)


def stripe_row_indices(height, rows_per_block, rank, n_ranks):
    n = stripe_rows(height, rows_per_block, rank, n_ranks)
    return np.array([_hip.hrt_stripe_row_index(height, rows_per_block, rank, n_ranks, i) for i in range(n)], dtype=np.int64)


class DeviceScene:
    """hrt_scene: the flat scene resident on one GPU."""

    def __init__(self, flat, device=0):
        self._h = None
        h = _vp()
        flat_ptr = flat if isinstance(flat, C.POINTER(FlatScene)) else C.pointer(flat)
        _check(_hip.hrt_scene_create(flat_ptr, device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            _hip.hrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_tile(self, cam, params, rect=None):
        """-> (h, w, 3) fp32 linear film tile, Stats."""
        if rect is None:
            rect = Rect(0, 0, params.width, params.height)
        elif not isinstance(rect, Rect):
            rect = Rect(*rect)
        out = np.empty((rect.h, rect.w, 3), dtype=np.float32)
        st = Stats()
        _check(_hip.hrt_render_tile(self._h, C.byref(cam), C.byref(params), rect, _ptr(out), C.byref(st)))
        return out, st

    def progress(self):
        """(paths ended, paths of the call) of the render call with FLAG_PROGRESS that is running, or ran last, on this scene
        (hrt_scene_progress: a plain read of host-mapped memory, callable from another thread while the render runs)."""
        d, t = C.c_uint64(0), C.c_uint64(0)
        _hip.hrt_scene_progress.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _check(_hip.hrt_scene_progress(self._h, C.byref(d), C.byref(t)))
        return d.value, t.value

    def render_stripes(self, cam, params, rows_per_block, rank, n_ranks):
        rows = stripe_rows(params.height, rows_per_block, rank, n_ranks)
        out = np.empty((rows, params.width, 3), dtype=np.float32)
        st = Stats()
        _check(_hip.hrt_render_stripes(self._h, C.byref(cam), C.byref(params), rows_per_block, rank, n_ranks, _ptr(out), C.byref(st)))
        return out, st

    def render_stripes_device(self, cam, params, rows_per_block, rank, n_ranks, d_out_ptr, stream=0):
        """Asynchronous: d_out_ptr is a device pointer (e.g. torch tensor .data_ptr()), stream a hipStream_t value."""
        _check(_hip.hrt_render_stripes_device(self._h, C.byref(cam), C.byref(params), rows_per_block, rank, n_ranks,
                                              _vp(d_out_ptr), _vp(stream)))

    def render_stripes_accumulate(self, cam, params, rows_per_block, rank, n_ranks, accum, sample_first, sample_count):
        """Progressive pass: adds samples [sample_first, sample_first + sample_count) to the host array `accum`
        (rows x W x 3 float32, running sums; divided by params.samples by the pass that reaches it)."""
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        st = Stats()
        _check(_hip.hrt_render_stripes_accumulate(self._h, C.byref(cam), C.byref(params), rows_per_block, rank, n_ranks, _ptr(accum),
                                                  sample_first, sample_count, C.byref(st)))
        return st

    def render_stripes_accumulate_device(self, cam, params, rows_per_block, rank, n_ranks, d_accum_ptr, sample_first, sample_count, stream=0):
        _check(_hip.hrt_render_stripes_accumulate_device(self._h, C.byref(cam), C.byref(params), rows_per_block, rank, n_ranks,
                                                         _vp(d_accum_ptr), sample_first, sample_count, _vp(stream)))

    def stats(self):
        st = Stats()
        _check(_hip.hrt_scene_stats(self._h, C.byref(st)))
        return st

    def resolve_u8(self, rgb_linear):
        a = _f32(rgb_linear)
        out = np.empty(a.shape, dtype=np.uint8)
        _check(_hip.hrt_resolve_u8(self._h, _ptr(a), a.size // 3, _ptr(out, _u8p)))
        return out

    def resolve_u8_device(self, d_in_ptr, n_pixels, d_out_ptr, stream=0):
        _check(_hip.hrt_resolve_u8_device(self._h, _vp(d_in_ptr), n_pixels, _vp(d_out_ptr), _vp(stream)))

    def closest_hit(self, params, origins, dirs, t_min=0.001, t_max=float("inf"), pixel0=0):
        o = _f32(origins)
        d = _f32(dirs)
        n = o.shape[0]
        out = np.zeros(n, dtype=HIT_DTYPE)
        _check(_hip.hrt_closest_hit(self._h, C.byref(params), n, _ptr(o), _ptr(d), t_min, t_max, pixel0,
                                    out.ctypes.data_as(C.POINTER(Hit))))
        return out


class MultiScene:
    """hrt_multi_*: the flat scene on several devices of this process + the RCCL gather of their film stripes."""

    def __init__(self, flat, devices=(0,), force_rccl=False, loopback=False):
        """loopback: the test mode of hrt_multi_create (force_rccl < 0) -- a device may be listed once per logical rank and the
        gather is one device copy per rank instead of ncclAllGather."""
        flat_ptr = flat if not isinstance(flat, FlatScene) else C.pointer(flat)
        devs = (C.c_int32 * len(devices))(*devices)
        h = _vp()
        _check(_hip.hrt_multi_create(flat_ptr, len(devices), devs, -1 if loopback else (1 if force_rccl else 0), C.byref(h)))
        self._h = h
        self._keep = flat

    def close(self):
        if getattr(self, "_h", None):
            _hip.hrt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def progress(self):
        d, t = C.c_uint64(0), C.c_uint64(0)
        _hip.hrt_multi_progress.argtypes = [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _check(_hip.hrt_multi_progress(self._h, C.byref(d), C.byref(t)))
        return d.value, t.value

    @property
    def uses_rccl(self):
        return bool(_hip.hrt_multi_uses_rccl(self._h))

    def render(self, cam, params, rows_per_block=8, sample_first=0, sample_count=-1, resume_sums=None, want_u8=True):
        """-> (sums or means [H, W, 3] float32, u8 film [H, W, 3] or None, Stats)"""
        sums = np.empty((params.height, params.width, 3), dtype=np.float32)
        u8 = np.empty((params.height, params.width, 3), dtype=np.uint8) if want_u8 else None
        rs = None if resume_sums is None else _f32(resume_sums)
        st = Stats()
        _check(_hip.hrt_multi_render(self._h, C.byref(cam), C.byref(params), rows_per_block, sample_first, sample_count,
                                     _ptr(rs) if rs is not None else None, _ptr(sums), _ptr(u8, _u8p) if want_u8 else None, C.byref(st)))
        return sums, u8, st


def math_probe(op, a, b=None, device=0):
    a = _f32(a)
    if op == 5:
        n = a.size // 4
        out = np.empty(n * 4, dtype=np.float32)
    else:
        n = a.size
        out = np.empty(n, dtype=np.float32)
    bb = _f32(b) if b is not None else None
    _check(_hip.hrt_math_probe(device, op, n, _ptr(a), _ptr(bb) if bb is not None else None, _ptr(out)))
    return out


def debug_bounds_violations(device=0):
    """The 8 violation counters of a -DHRT_DEBUG_BOUNDS build (read and cleared), or None from a normal build."""
    out = (C.c_int64 * 8)()
    _hip.hrt_debug_bounds_violations.argtypes = [C.c_int32, C.POINTER(C.c_int64)]
    st = _hip.hrt_debug_bounds_violations(device, out)
    if st == HRT_ERR_UNSUPPORTED:
        return None
    _check(st)
    return list(out)


def version():
    return _hip.hrt_version().decode()
