"""The C-ABI libraries load and export every symbol include/*.h declares, and the ctypes mirrors in
hobbyraytracer_amd/api.py have the layouts a C compiler gives the header structs.  No compute calls
(there is no GPU in the CPU test run)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(hrt_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(n for n in names if n not in ("hrt_status",)))


def test_hip_library_exports_every_declared_symbol(built):
    from hobbyraytracer_amd import api
    lib = C.CDLL(api.HIP_LIB_PATH)
    declared = _declared_functions("hrt.h")
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"libhrt_hip.so does not export {name}"
    assert sorted(api.HIP_SYMBOLS) == declared


def test_host_library_exports_every_declared_symbol(built):
    from hobbyraytracer_amd import api
    lib = C.CDLL(api.HOST_LIB_PATH)
    declared = _declared_functions("hrt_host.h")
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"libhrt_host.so does not export {name}"
    assert sorted(api.HOST_SYMBOLS) == declared


def test_struct_layouts_match_the_header(built, tmp_path):
    from hobbyraytracer_amd import api
    structs = {"hrt_xform": api.Xform, "hrt_prim": api.Prim, "hrt_matvec3": api.MatVec3, "hrt_matscalar": api.MatScalar,
               "hrt_material": api.Material, "hrt_texture": api.Texture, "hrt_mesh": api.Mesh, "hrt_bvh_node": api.BvhNode,
               "hrt_flat_scene": api.FlatScene, "hrt_camera": api.Camera, "hrt_params": api.Params, "hrt_rect": api.Rect,
               "hrt_stats": api.Stats, "hrt_hit": api.Hit}
    src = tmp_path / "sz.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/hrt.h"', 'int main(void){']
    for n in structs:
        lines.append(f'printf("{n} %zu\\n", sizeof({n}));')
    lines += ['printf("off_flat_nodes %zu\\n", offsetof(hrt_flat_scene, nodes));',
              'printf("off_flat_bg %zu\\n", offsetof(hrt_flat_scene, background_tex));',
              'printf("off_tex_offset %zu\\n", offsetof(hrt_texture, offset));',
              'printf("off_prim_xf %zu\\n", offsetof(hrt_prim, xf));', 'return 0;}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for n, t in structs.items():
        assert int(out[n]) == C.sizeof(t), n
    assert int(out["off_flat_nodes"]) == api.FlatScene.nodes.offset
    assert int(out["off_flat_bg"]) == api.FlatScene.background_tex.offset
    assert int(out["off_tex_offset"]) == api.Texture.offset.offset
    assert int(out["off_prim_xf"]) == api.Prim.xf.offset
    assert C.sizeof(api.BvhNode) == 64


def test_headers_compile_as_plain_c(built, tmp_path):
    """The boundary is a C ABI: both headers must be valid C99 with no C++ or torch types."""
    src = tmp_path / "c.c"
    src.write_text(f'#include "{ROOT}/include/hrt_host.h"\nint main(void){{ hrt_params p; (void)p; return 0; }}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-c", "-o", str(tmp_path / "c.o"), str(src)])


def test_status_strings_and_stripe_partition_need_no_gpu(built):
    from hobbyraytracer_amd import api
    assert "gfx950" in api.version()
    # partition: every row exactly once, blocks interleaved
    for H, R, G in [(640, 8, 8), (1080, 8, 4), (37, 5, 3), (16, 8, 1), (7, 8, 2)]:
        seen = []
        for r in range(G):
            idx = api.stripe_row_indices(H, R, r, G)
            assert len(idx) == api.stripe_rows(H, R, r, G)
            for i in idx:
                assert (i // R) % G == r
            seen += idx.tolist()
        assert sorted(seen) == list(range(H))


def test_scene_validation_runs_before_any_device_is_touched(built, tmp_path):
    """hrt_scene_create validates the flat scene first, so malformed input is refused with HRT_ERR_INVALID / UNSUPPORTED on
    a box without a GPU too (a valid scene gets HRT_ERR_NO_DEVICE there, HRT_OK on a GPU box).  Cases found by
    tests/tools/fuzz_flat.py: a negative texture width passed (only width > 0 was checked) and indexed in front of the texel
    array; offset + size could wrap around 2^64."""
    import ctypes as C
    from hobbyraytracer_amd import api
    api.write_hall_hdr(str(tmp_path / "old_hall_4k.hdr"), 32, 16)
    api.write_teapot_obj(str(tmp_path / "teapot.obj"), 0.05)
    hs = api.HostScene(os.path.join(os.path.dirname(__file__), "golden", "scenes", "teapot_scene.yaml"), str(tmp_path))

    def create(flat):
        h = C.c_void_p()
        st = api._hip.hrt_scene_create(C.byref(flat), 0, C.byref(h))
        if st == api.HRT_OK:
            api._hip.hrt_scene_destroy(h)
        return st, api._hip.hrt_last_error().decode()

    def with_texture(edit):
        flat = api.FlatScene()
        C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
        texs = (api.Texture * flat.n_textures)()
        C.memmove(texs, flat.textures, C.sizeof(texs))
        env = [i for i in range(flat.n_textures) if texs[i].kind == api.TEX_ENV][0]
        edit(texs[env])
        flat.textures = C.cast(texs, C.POINTER(api.Texture))
        return create(flat) + (texs,)

    assert create(hs.flat)[0] in (api.HRT_OK, api.HRT_ERR_NO_DEVICE)
    for edit, what in ((lambda t: setattr(t, "width", -1), "negative texture size"), (lambda t: setattr(t, "height", -7), "negative texture size"),
                       (lambda t: setattr(t, "offset", 2**64 - 8), "env texels out of range"), (lambda t: setattr(t, "width", 2**31 - 1), "env texels out of range"),
                       (lambda t: setattr(t, "channels", 2), "environment map needs >= 3 channels"), (lambda t: setattr(t, "channels", 2**31 - 1), "env texels out of range")):
        st, msg, _ = with_texture(edit)
        assert st == api.HRT_ERR_INVALID and what in msg, (st, msg)
    st, msg, _ = with_texture(lambda t: setattr(t, "width", 0))        # the reference's "failed to load" texture: allowed (cyan)
    assert st in (api.HRT_OK, api.HRT_ERR_NO_DEVICE), msg


def test_cli_binary_exists_and_reports_load_failure(built, tmp_path):
    from hobbyraytracer_amd import api
    assert os.access(api.CLI_PATH, os.X_OK)
    # main.cpp:155-156: a scene that fails to load makes the process return -1 (exit status 255)
    p = subprocess.run([api.CLI_PATH, str(tmp_path / "missing.yaml")], capture_output=True, cwd=tmp_path)
    assert p.returncode == 255


def test_product_package_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under hobbyraytracer_amd/ or include/ may name it."""
    bad = []
    for base in ("hobbyraytracer_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    # comments may MENTION the oracle; what is forbidden is loading, importing or including it
                    if re.search(r"liboracle|oracle_py|import oracle|from oracle|#include\s*[\"<][^\">]*oracle|dlopen\([^)]*oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_stripe_partition_and_its_inverse_with_more_ranks_than_row_blocks(built):
    """hrt_stripe_rows / hrt_stripe_row_index (the multi-GPU row partition) against the inverse map the gather kernels of
    hrt_multi_render use (k_unstripe: block b = row / R belongs to rank b % G, local row (b / G) * R + row % R), including
    partitions in which some ranks own nothing (G > number of row blocks: a 48-row film on 8 GPUs has 6 blocks of 8 rows)."""
    import numpy as np
    from hobbyraytracer_amd import api
    r = np.random.default_rng(5)
    cases = [(48, 8, 8), (5, 8, 4), (77, 8, 1), (640, 8, 8), (1080, 16, 3)] + [(int(r.integers(2, 300)), int(r.integers(1, 20)), int(r.integers(1, 12))) for _ in range(40)]
    for H, R, G in cases:
        rows = [api.stripe_rows(H, R, g, G) for g in range(G)]
        assert sum(rows) == H and rows[0] == max(rows)                       # rank 0 owns the largest share (what the padded gather is sized by)
        if G > (H + R - 1) // R:
            assert rows[-1] == 0                                             # ranks without a block
        seen = np.zeros(H, bool)
        for g in range(G):
            idx = api.stripe_row_indices(H, R, g, G)
            assert len(idx) == rows[g] and (np.diff(idx) > 0).all()
            for local, row in enumerate(idx):
                b = row // R
                assert b % G == g and (b // G) * R + (row - b * R) == local    # k_unstripe / k_restripe's index arithmetic
                assert not seen[row]
                seen[row] = True
            assert api.stripe_row_index(H, R, g, G, rows[g]) == -1
        assert seen.all()


def test_scene_create_refuses_bad_reference_order_codes_and_deep_checkers_without_a_gpu(built, assets, scenes_dir):
    """hrt_scene_create validates before it touches a device (so this runs on the CPU-only box: a good scene gets as far as
    HRT_ERR_NO_DEVICE): tri_ref_order must be the depth-first code of a median-split tree -- the reference's own BVH is derived
    from it for the rays that walk it (hrt_pack.h pack_ref_tree) --, and checkered textures may nest 4 deep at most."""
    import copy
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    def status_of(flat):
        try:
            api.DeviceScene(flat, 0).close()
            return api.HRT_OK
        except api.HrtError as e:
            return e.status
    good = status_of(hs.flat_ptr)
    if good == api.HRT_OK:
        pytest.skip("a GPU is present: the no-device ordering is what this test is about")
    assert good == api.HRT_ERR_NO_DEVICE
    flat = api.FlatScene()
    C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
    n = flat.n_tris
    codes = (C.c_uint32 * n)(*[flat.tri_ref_order[i] for i in range(n)])
    codes[7] = 12345678
    flat.tri_ref_order = C.cast(codes, C.POINTER(C.c_uint32))
    assert status_of(flat) == api.HRT_ERR_INVALID
    codes[7] = codes[8]                                                  # a code used twice
    assert status_of(flat) == api.HRT_ERR_INVALID
    # a cyclic checker
    C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
    nt = flat.n_textures
    texs = (api.Texture * (nt + 1))(*[flat.textures[i] for i in range(nt)])
    texs[nt].kind = api.TEX_CHECKER; texs[nt].even = nt; texs[nt].odd = 0
    flat.textures = C.cast(texs, C.POINTER(api.Texture)); flat.n_textures = nt + 1
    assert status_of(flat) == api.HRT_ERR_UNSUPPORTED
