"""GPU parity over every scene class (BASELINE configs C2-C5 at reduced film sizes) and size-independent
properties at BASELINE's full film sizes.  All through the C ABI; expected result: bit equality with the
oracle (tolerance written where a comparison is made)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = {  # name -> (W, H, spp)
    "shiny_teapot.yaml": (160, 90, 16),     # C4: metal roughness 0.2 + env map, bare mesh (0 wrappers, Q-3)
    "cornell_box.yaml": (96, 96, 16),       # C2: rects, boxes (rotate_y / rotate wrappers), rough dielectric + metal spheres
    "bust_scene.yaml": (80, 80, 8),         # C5: dielectric mesh in one Translate, ConstantMedium (RNG inside hit), checker floor
    "material_zoo.yaml": (96, 96, 12),      # image / checker textures, pbr, uv_test, isotropic in a box, 3-wrapper mesh
    "triangles.yaml": (96, 64, 8),          # the stand-alone Triangle class (triangle.cpp:4-40), also under wrappers and as a light
    "three_meshes.yaml": (96, 64, 8),       # three meshes (3 wrappers / 1 / none) interleaved with spheres and rects: k_wf_pre, one traversal per mesh
}


# the three ways a film gets rendered: per-round kernels (k_wf_gen / pre / ext / shade, what large batches use), every round of
# a task inside k_wf_tail (what these small films use by default), the megakernel
@pytest.mark.parametrize("path", ["wavefront-rounds", "wavefront-tail", "megakernel"])
@pytest.mark.parametrize("quirks", ["reference", "fixed"])
@pytest.mark.parametrize("scene", sorted(SCENES))
def test_image_parity(built, assets, scenes_dir, scene, quirks, path, monkeypatch):
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    monkeypatch.setenv("HRT_WF_TAIL_ROUND", "1000" if path == "wavefront-rounds" else "1")
    W, H, spp = SCENES[scene]
    hs = api.HostScene(f"{scenes_dir}/{scene}", assets)
    q = api.QUIRKS_REFERENCE if quirks == "reference" else api.QUIRKS_FIXED
    params = api.default_params(W, H, spp, quirks=q, stats=True, megakernel=(path == "megakernel"))
    cam = hs.camera(W, H)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    img, st = dev.render_tile(cam, params)
    ref, sr = orc.World(hs.flat_ptr).render_tile(cam, params)
    assert (st.rays, st.samples, st.mesh_hits, st.env_lookups) == (sr.rays, sr.samples, sr.mesh_hits, sr.env_lookups)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(img), finite)
    # stated tolerance: 1e-6 relative per fp32 film value; observed: identical bits
    np.testing.assert_allclose(img[finite], ref[finite], rtol=1e-6, atol=1e-7)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(dev.resolve_u8(img), orc.resolve_u8(ref))
    dev.close()


def test_closest_hit_all_primitive_kinds(built, assets, scenes_dir):
    """hitRecord parity (t, p, normal, u, v, frontFace) for spheres, rects, boxes under RotateY / RotateQuat,
    a medium (RNG inside hit), a 3-wrapper mesh, stand-alone Triangles, several meshes in one world."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    r = np.random.default_rng(17)
    for scene in ("cornell_box.yaml", "material_zoo.yaml", "bust_scene.yaml", "triangles.yaml", "three_meshes.yaml"):
        hs = api.HostScene(f"{scenes_dir}/{scene}", assets)
        dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
        o = r.uniform([-3, 0.05, -3], [3, 4.5, 7], (150000, 3)).astype(np.float32)
        d = (r.uniform([-2.5, 0, -2.5], [2.5, 3.5, 2.5], (150000, 3)) - o).astype(np.float32)
        for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
            p = api.default_params(16, 16, 1, quirks=q, seed=5)
            g, c = dev.closest_hit(p, o, d, pixel0=1000), world.closest_hit(p, o, d, pixel0=1000)
            same = (g["prim"] == c["prim"]) & (g["tri"] == c["tri"])
            # both quirk sets: with Q-4 the rays whose direction all but vanishes on the origin-chosen shear axis (triangle.cpp:70)
            # walk the reference's own tree (hrt_device.h ref_walk), so even their meaningless t comes out as the reference's
            assert same.all(), (scene, q, np.nonzero(~same)[0][:5])
            hit = (c["prim"] >= 0) & same
            assert len(np.unique(c["prim"][hit])) >= min(3, hs.flat.n_prims - 1)
            for f in ("t", "p", "normal", "u", "v"):
                assert np.array_equal(g[f][hit].view(np.uint32), c[f][hit].view(np.uint32)), (scene, f)
            assert np.array_equal(g["front_face"][hit], c["front_face"][hit])
        dev.close()


def test_render_is_deterministic_and_seed_sensitive(built, assets, scenes_dir):
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    cam = hs.camera(128, 128)
    a, sa = dev.render_tile(cam, api.default_params(128, 128, 8, seed=1))
    b, sb = dev.render_tile(cam, api.default_params(128, 128, 8, seed=1))
    c, _ = dev.render_tile(cam, api.default_params(128, 128, 8, seed=2))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.rays == sb.rays
    assert not np.array_equal(a, c)
    # two seeds estimate the same image: means agree within Monte-Carlo noise (3 %)
    np.testing.assert_allclose(a.mean((0, 1)), c.mean((0, 1)), rtol=0.03)
    dev.close()


def test_full_size_properties_teapot_640(built, assets, scenes_dir):
    """BASELINE's headline film (640x640, 100 spp) through size-independent properties: sample count,
    stripes == full frame bit for bit, the STATS build changes no pixel, every value finite and >= 0,
    and the top-left 64x8 block equals the oracle."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    W = H = 640
    cam = hs.camera(W, H)
    p = api.default_params(W, H, 100)
    full, st = dev.render_tile(cam, p)
    assert st.samples == W * H * 100 and st.rays > st.samples and st.launches == 1
    assert np.isfinite(full).all() and (full >= 0).all()
    full_s, st_s = dev.render_tile(cam, api.default_params(W, H, 100, stats=True))
    assert np.array_equal(full.view(np.uint32), full_s.view(np.uint32)) and st_s.rays == st.rays and st_s.box_tests > 0
    out = np.zeros_like(full)
    for rank in range(8):
        part, _ = dev.render_stripes(cam, p, 8, rank, 8)
        out[api.stripe_row_indices(H, 8, rank, 8)] = part
    assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
    ref, _ = orc.World(hs.flat_ptr).render_tile(cam, p, (0, 0, 64, 8))
    assert np.array_equal(full[:8, :64].view(np.uint32), ref.view(np.uint32))
    dev.close()


def test_scene_validation_refuses_malformed_input(built, assets, scenes_dir):
    """hrt_scene_create validates every index the kernels follow (a malformed scene must not reach the GPU)."""
    import copy
    import ctypes as C
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    flat = api.FlatScene()
    C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
    flat.background_tex = 99
    with pytest.raises(api.HrtError) as e:
        api.DeviceScene(flat, 0)
    assert e.value.status == api.HRT_ERR_INVALID
    C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
    nodes = (api.BvhNode * flat.n_nodes)()
    C.memmove(nodes, flat.nodes, C.sizeof(nodes))
    nodes[0].child0 = 0          # a cycle: the root points at itself
    flat.nodes = C.cast(nodes, C.POINTER(api.BvhNode))
    with pytest.raises(api.HrtError):
        api.DeviceScene(flat, 0)
    with pytest.raises(api.HrtError):
        api.DeviceScene(hs.flat_ptr, 0).render_tile(hs.camera(8, 8), api.default_params(8, 8, 1), (4, 4, 8, 8))   # tile outside the film
    with pytest.raises(api.HrtError):
        api.DeviceScene(hs.flat_ptr, 77)                                                                       # no such device
    dev = api.DeviceScene(hs.flat_ptr, 0)
    for bad in (dict(max_depth=0), dict(max_depth=10**6)):                                                     # rounds: 1 .. 65536
        with pytest.raises(api.HrtError):
            dev.render_tile(hs.camera(8, 8), api.default_params(8, 8, 1, **bad))
    nan_tmin = api.default_params(8, 8, 1); nan_tmin.t_min = float("nan")
    zero_spp = api.default_params(8, 8, 1); zero_spp.samples = 0
    tiny = api.default_params(8, 8, 1); tiny.width = 1
    for p in (nan_tmin, zero_spp, tiny):
        with pytest.raises(api.HrtError):
            dev.render_tile(hs.camera(8, 8), p)
    dev.close()
    C.memmove(C.byref(flat), hs.flat_ptr, C.sizeof(api.FlatScene))
    pos = (C.c_float * (flat.n_tris * 9))()
    C.memmove(pos, flat.tri_pos, C.sizeof(pos))
    pos[9 * 100 + 4] = float("nan")                                                                            # a NaN vertex coordinate
    flat.tri_pos = C.cast(pos, C.POINTER(C.c_float))
    with pytest.raises(api.HrtError) as e:
        api.DeviceScene(flat, 0)
    assert e.value.status == api.HRT_ERR_INVALID and "non-finite vertex position (triangle 100)" in str(e.value)
    pos[9 * 100 + 4] = 1e38                                                                                    # finite, but the mesh extent could overflow fp32
    with pytest.raises(api.HrtError) as e:
        api.DeviceScene(flat, 0)
    assert e.value.status == api.HRT_ERR_UNSUPPORTED and "beyond +-8e37 (triangle 100)" in str(e.value)


def test_device_pointer_stream_api_with_torch(built, assets, scenes_dir):
    """hrt_render_stripes_device / hrt_resolve_u8_device on a caller-owned device buffer and HIP stream (the form
    bench.py and a one-process-per-GPU host use): results equal the host-buffer entry points bit for bit."""
    import torch
    from hobbyraytracer_amd import api, tiles
    hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    W, H, spp, R = 96, 72, 6, 8
    cam, p = hs.camera(W, H), api.default_params(W, H, spp)
    full, _ = dev.render_tile(cam, p)
    stream = torch.cuda.Stream()
    film = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    with torch.cuda.stream(stream):
        for G in (1, 3):
            for rank in range(G):
                layout = tiles.StripeLayout(H, W, R, G)
                tile = torch.full((layout.max_rows, W, 3), float("nan"), dtype=torch.float32, device="cuda")
                dev.render_stripes_device(cam, p, R, rank, G, tile.data_ptr(), stream.cuda_stream)
                rows = torch.as_tensor(layout.row_indices(rank), device="cuda")
                film[rows] = tile[:len(rows)]
            stream.synchronize()
            assert np.array_equal(film.cpu().numpy().view(np.uint32), full.view(np.uint32)), f"G={G}"
        u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        dev.resolve_u8_device(film.data_ptr(), W * H, u8.data_ptr(), stream.cuda_stream)
        stream.synchronize()
    assert np.array_equal(u8.cpu().numpy(), dev.resolve_u8(full))
    st = dev.stats()
    assert st.launches == 4 and st.samples == W * H * spp * 2      # 1 + 3 stripe launches cover the film twice
    dev.close()


def test_world_larger_than_the_lds_tables(built, assets, tmp_path):
    """150 spheres with 40 materials + a mesh: prims/materials no longer fit the 12 KB LDS staging area of the
    kernels (hrt_device.h stage_tables), which then read them from global memory through the same pointers.
    Same film as the oracle, both paths."""
    import shutil
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    r = np.random.default_rng(12)
    mats, objs = [], []
    for i in range(40):
        c = r.uniform(0.1, 0.9, 3)
        kind = ("lambertian", "metal", "dielectric")[i % 3]
        extra = {"lambertian": "", "metal": f"    roughness: {r.uniform(0, 0.5):.3f}\n", "dielectric": "    ior: 1.5\n"}[kind]
        mats.append(f"  - name: m{i}\n    type: {kind}\n    albedo: [{c[0]:.3f}, {c[1]:.3f}, {c[2]:.3f}]\n{extra}")
    for i in range(150):
        x, z = r.uniform(-6, 6), r.uniform(-6, 4)
        rad = r.uniform(0.15, 0.35)
        objs.append(f"  - type: sphere\n    center: [{x:.3f}, {rad:.3f}, {z:.3f}]\n    radius: {rad:.3f}\n    material: m{i % 40}\n")
    yaml = ("film:\n    width: 96\n    height: 64\n    samples: 6\n    output: many.png\n"
            "camera:\n    position: [0, 2.5, 9]\n    look_at: [0, 0.5, 0]\n    up: [0, 1, 0]\n    fov: 40\n    aperture: 0\n    focal_distance: 9\n"
            "    background: [0.6, 0.7, 0.9]\n"
            "materials:\n  - name: ground\n    type: lambertian\n    albedo: [0.5, 0.5, 0.5]\n" + "".join(mats) +
            "objects:\n  - type: xz_rect\n    x: [-8, 8]\n    z: [-8, 8]\n    k: 0\n    material: ground\n" + "".join(objs[:75]) +
            "  - type: mesh\n    path: teapot.obj\n    material: m1\n    transform:\n        translate: [0, 0, -1]\n" + "".join(objs[75:]))
    (tmp_path / "many.yaml").write_text(yaml)
    shutil.copy(f"{assets}/teapot.obj", tmp_path / "teapot.obj")
    hs = api.HostScene(str(tmp_path / "many.yaml"), str(tmp_path))
    assert hs.flat.n_prims == 152 and hs.flat.n_prims * 140 > 12288
    dev = api.DeviceScene(hs.flat_ptr, 0)
    W, H, spp = 96, 64, 6
    cam = hs.camera(W, H)
    ref, sref = orc.World(hs.flat_ptr).render_tile(cam, api.default_params(W, H, spp, stats=True))
    for mega in (False, True):
        img, st = dev.render_tile(cam, api.default_params(W, H, spp, stats=True, megakernel=mega))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"megakernel={mega}"
        assert (st.rays, st.samples, st.mesh_hits, st.env_lookups) == (sref.rays, sref.samples, sref.mesh_hits, sref.env_lookups)
    dev.close()


from tests.scene_helpers import chain_scene as _chain_scene  # noqa: E402


@pytest.mark.parametrize("n,lo,hi", [(90, 21, 24), (100, 25, 32)])
def test_deep_bvh_uses_the_larger_stack_variants(built, tmp_path, monkeypatch, n, lo, hi):
    """BVH depth 21..24 -> the 24-entry LDS stack kernels, 25..32 -> the 32-entry ones (k_wf_ext and k_wf_tail are compiled
    for 20 / 24 / 32 entries and chosen per mesh): same film and hit records as the oracle on all three render paths."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    hs = api.HostScene(_chain_scene(tmp_path, n, 1.5), str(tmp_path))
    assert lo <= hs.bvh_depth(0) <= hi
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W, H, spp = 64, 48, 4
    cam = hs.camera(W, H)
    ref, sr = world.render_tile(cam, api.default_params(W, H, spp, stats=True))
    assert sr.mesh_hits > 500
    for tail, mega in (("1", False), ("1000", False), ("1", True)):
        monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
        img, st = dev.render_tile(cam, api.default_params(W, H, spp, stats=True, megakernel=mega))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (tail, mega)
        assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits)
    r = np.random.default_rng(3)
    o = r.uniform([0, -0.5, 2], [40, 3, 14], (50000, 3)).astype(np.float32)
    d = (r.uniform([0, -0.5, -0.2], [40, 0.5, 0.2], (50000, 3)) - o).astype(np.float32)
    p = api.default_params(8, 8, 1, quirks=api.QUIRKS_FIXED)
    g, c = dev.closest_hit(p, o, d), world.closest_hit(p, o, d)
    assert (c["tri"] >= 0).sum() > 2000 and np.array_equal(g["tri"], c["tri"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    dev.close()


@pytest.mark.parametrize("kind", ["dup", "degenerate"])
def test_triangle_soup_ties_and_zero_area_faces(built, tmp_path, monkeypatch, kind):
    """Duplicated faces whose copies carry opposite normals (every hit on them is an exact tie: the reference's walk order
    decides, hrt_device.h trav_leaf "Ties") and zero-area faces: film and hit records as the oracle, on all three paths."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import soup_scene
    path, ctr = soup_scene(tmp_path, kind)
    hs = api.HostScene(path, str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W = H = 48
    cam = hs.camera(W, H)
    for quirks in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(W, H, 4, quirks=quirks, stats=True))
        assert sr.mesh_hits > 300
        for tail, mega in (("1", False), ("1000", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(W, H, 4, quirks=quirks, stats=True, megakernel=mega))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (quirks, tail, mega)
            assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits)
    r = np.random.default_rng(3)
    n = 50000
    tgt = ctr + r.uniform(-1.2, 1.2, (n, 3))
    dirs = r.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o, d = (tgt + dirs * 4).astype(np.float32), (-dirs).astype(np.float32)
    p = api.default_params(8, 8, 1, quirks=api.QUIRKS_FIXED)
    g, c = dev.closest_hit(p, o, d), world.closest_hit(p, o, d)
    assert (c["tri"] >= 0).sum() > 10000 and np.array_equal(g["tri"], c["tri"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    assert np.array_equal(g["normal"].view(np.uint32), c["normal"].view(np.uint32))
    dev.close()


@pytest.mark.parametrize("meshes", [False, True])
@pytest.mark.parametrize("extreme", [0, 1])
def test_random_worlds(built, tmp_path, monkeypatch, extreme, meshes):
    """tests/scene_helpers.py random_world (coincident surfaces, duplicated objects, transform chains; `extreme`: zero /
    negative radii and dimensions, reversed rect ranges, ior 1e-3..50, densities 0 / 1e4 / -1 ...): 24 seeds x 2 quirk sets,
    wavefront pipeline (with and without the task-persistent tail) and megakernel, film bit-identical to the oracle.  `meshes`
    adds 1..3 triangle meshes (soup / smooth sphere / lattice-aligned grid) anywhere in the object list, so the pipeline's
    mesh-by-mesh walk with analytic primitives in between (k_wf_pre) is exercised."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import random_world, films_equal
    rendered = 0
    for seed in range(24):
        hs = api.HostScene(random_world(tmp_path, 1000 * extreme + seed, extreme, meshes=meshes, images=meshes), str(tmp_path))
        dev = api.DeviceScene(hs.flat_ptr, 0)
        rendered += 1
        world = orc.World(hs.flat_ptr)
        cam = hs.camera(40, 40)
        for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
            ref, sr = world.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=True))
            for tail, mega in (("1", False), ("1000", False), ("1", True)):
                monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
                img, st = dev.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=True, megakernel=mega))
                assert st.rays == sr.rays, (seed, q, tail, mega)
                assert films_equal(img, ref), (seed, q, tail, mega)
        dev.close()
    assert rendered == 24


@pytest.mark.parametrize("chain", ["", "Y", "Q", "S", "T", "YQ", "QS", "ST", "YQS", "YQT", "QST", "YST", "YQST"])
def test_wrapper_chains_of_every_length(built, tmp_path, monkeypatch, chain):
    """Every primitive kind under wrapper chains of length 0..4.  Pins a compiler problem found by the random worlds:
    world_rec's chain walk, written as four independent predicated blocks, came back with a wrong rec.p for chains of
    exactly three wrappers inside k_wf_shade / k_wf_tail only (hrt_device.h world_rec)."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import wrapper_chain_scene
    hs = api.HostScene(wrapper_chain_scene(tmp_path, chain), str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W = H = 48
    cam = hs.camera(W, H)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(W, H, 4, quirks=q, stats=True))
        for tail, mega, stats in (("1", False, True), ("1000", False, True), ("1000", False, False), ("1", False, False), ("1", True, True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(W, H, 4, quirks=q, stats=stats, megakernel=mega))
            assert st.rays == sr.rays, (q, tail, mega, stats)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (q, tail, mega, stats)
    dev.close()


def test_nan_rays_and_nan_t_max_from_a_degenerate_triangle(built, tmp_path, monkeypatch):
    """The GPU twin of the CPU test of the same name (tests/scene_helpers.py nan_ray_scene): a primitive that "hits" every ray
    with t = NaN in front of two meshes, on all three render paths."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import nan_ray_scene, films_equal
    hs = api.HostScene(nan_ray_scene(tmp_path), str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    cam = hs.camera(32, 32)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(32, 32, 4, quirks=q, stats=True))
        assert sr.mesh_hits > 1000
        for tail, mega in (("1", False), ("1000", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(32, 32, 4, quirks=q, stats=True, megakernel=mega))
            assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits), (q, tail, mega)
            assert films_equal(img, ref), (q, tail, mega)
    dev.close()


def test_wavefront_pipeline_keeps_the_nan_of_an_infinite_attenuation(built, tmp_path, monkeypatch):
    """random_world seed 14563 (extreme): a path whose attenuation has become infinite scatters on; main.cpp:66 adds
    attenuation * emitted = inf * 0 = NaN to the pixel at that bounce.  The pipeline's path state carries no running result
    (it is 0 while a path lives, for finite attenuations), so it reported -inf where the reference has NaN, in 15 % of this
    film's pixels; the NaN is now folded into the attenuation.  Found by tests/tools/gpu_fuzz.py once negative medium
    densities were let through."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import random_world, films_equal
    hs = api.HostScene(random_world(tmp_path, 14563, 1), str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    cam = hs.camera(40, 40)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=True))
        assert np.isnan(ref).any(2).sum() > 100
        for tail, mega in (("1", False), ("1000", False), ("3", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(40, 40, 4, quirks=q, stats=True, megakernel=mega))
            assert st.rays == sr.rays and films_equal(img, ref), (q, tail, mega)
    dev.close()


@pytest.mark.parametrize("n_mesh", [5, 9])
def test_many_mesh_instances_and_near_ties_on_shared_edges(built, tmp_path, monkeypatch, n_mesh):
    """The GPU twin of the CPU test of the same name: more meshes than k_wf_tail takes (the pipeline then never switches to
    it), 1-ulp near-ties on the teapot's shared edges; all render paths against the oracle."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import many_meshes_scene, films_equal
    hs = api.HostScene(many_meshes_scene(tmp_path, n_mesh), str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    cam = hs.camera(48, 48)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(48, 48, 4, quirks=q, stats=True))
        for tail, mega in (("1", False), ("1000", False), ("5", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(48, 48, 4, quirks=q, stats=True, megakernel=mega))
            assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits), (q, tail, mega)
            assert films_equal(img, ref), (q, tail, mega)
    dev.close()


def test_progressive_accumulation_equals_one_shot(built, assets, scenes_dir):
    """hrt_render_stripes_accumulate: any batching of the samples, with the accumulation buffer taken to the host
    (checkpoint) and brought back between passes, ends bit-identical to the one-shot render; previews are
    sums / samples_done; bad ranges and the megakernel with a partial range are refused."""
    from hobbyraytracer_amd import api
    hs = api.HostScene(f"{scenes_dir}/material_zoo.yaml", assets)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    W, H, spp, R = 72, 56, 16, 8
    cam, p = hs.camera(W, H), api.default_params(W, H, spp)
    full, _ = dev.render_tile(cam, p)
    for G, batches in ((1, (5, 7, 4)), (2, (1, 15)), (3, (16,))):
        film = np.zeros((H, W, 3), np.float32)
        for rank in range(G):
            rows = api.stripe_rows(H, R, rank, G)
            accum = np.full((rows, W, 3), np.nan, np.float32)      # sample_first == 0 must not read it
            s0 = 0
            for n in batches:
                st = dev.render_stripes_accumulate(cam, p, R, rank, G, accum, s0, n)
                assert st.samples == rows * W * n
                s0 += n
                if s0 < spp:   # preview of a partial render: the mean of the first s0 samples = a render at s0 spp, exactly
                    pp = api.default_params(W, H, s0)
                    part, _ = dev.render_stripes(cam, pp, R, rank, G)
                    assert np.array_equal(accum / np.float32(s0), part, equal_nan=True)
                accum = accum.copy()                                 # a checkpoint is just this array + s0
            idx = [api.stripe_row_index(H, R, rank, G, l) for l in range(rows)]
            film[idx] = accum
        assert np.array_equal(film.view(np.uint32), full.view(np.uint32)), f"G={G} batches={batches}"
    accum = np.zeros((H, W, 3), np.float32)
    for bad in ((-1, 4), (0, 0), (10, 7), (16, 1)):
        with pytest.raises(api.HrtError) as e:
            dev.render_stripes_accumulate(cam, p, R, 0, 1, accum, *bad)
        assert e.value.status == api.HRT_ERR_INVALID
    with pytest.raises(api.HrtError) as e:
        dev.render_stripes_accumulate(cam, api.default_params(W, H, spp, megakernel=True), R, 0, 1, accum, 0, 8)
    assert e.value.status == api.HRT_ERR_UNSUPPORTED
    dev.render_stripes_accumulate(cam, api.default_params(W, H, spp, megakernel=True), R, 0, 1, accum, 0, spp)   # the full range is fine
    assert np.array_equal(accum.view(np.uint32), full.view(np.uint32))
    dev.close()


FULL_SIZE = [  # BASELINE.json's configs as they stand: film size, spp and mesh size (C4 / C5 with the full-size stand-ins of `assets_full`)
    ("cornell_box.yaml", 640, 640, 256, "C2: 640x640x256, analytic primitives only"),
    ("teapot_scene.yaml", 1024, 1024, 256, "C3: 1024x1024x256 (268 M slots = 49 GB of wavefront state in one batch)"),
    ("shiny_teapot.yaml", 1920, 1080, 512, "C4: 1920x1080x512 = 1.06 G camera samples, 4096x2048 fp32 environment map"),
    ("bust_scene.yaml", 2048, 2048, 1024, "C5: 2048x2048x1024 = 2^32 camera samples, ~100k-triangle bust, rough dielectric + constant medium"),
]


@pytest.mark.parametrize("scene,W,H,spp,what", FULL_SIZE)
def test_full_size_properties(built, assets_full, scenes_dir, scene, W, H, spp, what):
    """BASELINE's configurations in full through size-independent properties: exact sample count (64-bit: C5 has 2^32), finite
    non-negative film, a 32x8 block equal to the oracle bit for bit at the configuration's own spp, and row-stripe rendering of
    two ranks equal to the full frame."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    quirks = api.QUIRKS_FIXED if scene == "teapot_scene.yaml" else api.QUIRKS_REFERENCE   # C3 = fixed quirks
    hs = api.HostScene(f"{scenes_dir}/{scene}", assets_full)
    if scene == "bust_scene.yaml":
        assert hs.flat.n_tris > 95_000 and max(hs.bvh_depth(m) for m in range(hs.flat.n_meshes)) >= 17   # ~100k triangles, a deep tree
        # (the 24- and 32-entry LDS stack variants of k_wf_ext: test_deep_bvh_uses_the_larger_stack_variants)
    dev = api.DeviceScene(hs.flat_ptr, 0)
    cam, p = hs.camera(W, H), api.default_params(W, H, spp, quirks=quirks)
    full, st = dev.render_tile(cam, p)
    assert st.samples == W * H * spp and st.rays >= st.samples
    finite = np.isfinite(full)
    assert finite.mean() > 0.9999 and (full[finite] >= 0).all()    # a NaN sample poisons only its own pixel (film.cpp:35-37 scrubs it)
    x0, y0 = W // 2 - 16, H // 2
    ref, _ = orc.World(hs.flat_ptr).render_tile(cam, p, (x0, y0, 32, 8))
    assert np.array_equal(full[y0:y0 + 8, x0:x0 + 32].view(np.uint32), ref.view(np.uint32)), what
    if W * H * spp <= 1_200_000_000:
        out = np.zeros_like(full)
        for rank in range(2):
            part, _ = dev.render_stripes(cam, p, 8, rank, 2)
            out[api.stripe_row_indices(H, 8, rank, 2)] = part
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
    dev.close()


def test_full_size_bust_hit_records_and_small_film(built, assets_full, scenes_dir, monkeypatch):
    """The ~100k-triangle bust of config C5 (binned-SAH tree: depth 18, the 20-entry LDS stack; the 24- and 32-entry variants run in
    test_deep_bvh_uses_the_larger_stack_variants and on this mesh's LBVH in test_device_bvh_builder_*): hitRecord parity of 150 000
    rays, and a small film on every render path, both quirk sets, against the oracle."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    hs = api.HostScene(f"{scenes_dir}/bust_scene.yaml", assets_full)
    assert hs.flat.n_tris > 95_000
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    r = np.random.default_rng(23)
    o = r.uniform([-3, 0.05, -3], [3, 4.5, 7], (150000, 3)).astype(np.float32)
    d = (r.uniform([-1.2, 0, -1.2], [1.2, 3.0, 1.2], (150000, 3)) - o).astype(np.float32)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        p = api.default_params(16, 16, 1, quirks=q, seed=5)
        g, c = dev.closest_hit(p, o, d, pixel0=1000), world.closest_hit(p, o, d, pixel0=1000)
        assert np.array_equal(g["prim"], c["prim"]) and np.array_equal(g["tri"], c["tri"]), q
        hit = c["prim"] >= 0
        assert (c["tri"] >= 0).sum() > 20000
        for f in ("t", "p", "normal", "u", "v"):
            assert np.array_equal(g[f][hit].view(np.uint32), c[f][hit].view(np.uint32)), (q, f)
        W, H, spp = 72, 72, 6
        cam = hs.camera(W, H)
        ref, sr = world.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True))
        for tail, mega in (("1", False), ("1000", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True, megakernel=mega))
            assert st.rays == sr.rays and sr.mesh_hits > 5000, (q, tail, mega)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (q, tail, mega)
    dev.close()


def test_bust_on_a_deep_tree_runs_the_32_entry_stack_kernels(built, assets_full, scenes_dir, monkeypatch):
    """BASELINE's C5 names a "deep-BVH LDS-stack stress": the host's SAH tree of the 100 k-triangle bust is 18 levels deep (the 20-entry
    stack), so the stress is made here -- the same mesh on its Morton LBVH from the GPU, 29 levels deep: k_wf_ext / k_wf_tail with the
    32-entry stack (4 instead of 6 blocks per CU) and the megakernel, both quirk sets, film and segment counts against the oracle."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    api.use_device_bvh_builder(True, algo="lbvh")
    try:
        hs = api.HostScene(f"{scenes_dir}/bust_scene.yaml", assets_full)
    finally:
        api.use_device_bvh_builder(False)
    depth = max(hs.bvh_depth(m) for m in range(hs.flat.n_meshes))
    assert hs.flat.n_tris > 95_000 and 25 <= depth <= 31, depth
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W, H, spp = 96, 96, 6
    cam = hs.camera(W, H)
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True))
        assert sr.mesh_hits > 8000
        for tail, mega in (("1000", False), ("2", False), ("1", True)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            img, st = dev.render_tile(cam, api.default_params(W, H, spp, quirks=q, stats=True, megakernel=mega))
            assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits), (q, tail, mega)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (q, tail, mega)
    dev.close()


def test_thin_lens_flag_on_every_render_path(built, assets, scenes_dir, monkeypatch):
    """HRT_FLAG_THIN_LENS (camera.h:34's commented-out circularRand(lensRadius), hrt.h hrt_camera): pipeline, tail and megakernel
    equal the oracle; the CPU twin (test_flat_vs_oracle_cpu.py) checks what the flag means."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    hs = api.HostScene(f"{scenes_dir}/bust_scene.yaml", assets)
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W, H, spp = 56, 40, 5
    cam = hs.camera(W, H)
    cam.lens_radius = 0.12
    ref, sr = world.render_tile(cam, api.default_params(W, H, spp, thin_lens=True, stats=True))
    pin, _ = world.render_tile(cam, api.default_params(W, H, spp))
    assert not np.array_equal(ref, pin)
    for tail, mega in (("1", False), ("1000", False), ("1", True)):
        monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
        img, st = dev.render_tile(cam, api.default_params(W, H, spp, thin_lens=True, stats=True, megakernel=mega))
        assert st.rays == sr.rays and np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (tail, mega)
    dev.close()


@pytest.mark.parametrize("enclosed", [True, False])
def test_wrapperless_glass_mesh_inherits_the_front_face_of_the_previous_object(built, tmp_path, monkeypatch, enclosed):
    """The GPU twin of the CPU test of the same name (tests/scene_helpers.py stale_front_face_scene): the stale
    hitRecord::frontFace of a mesh without a wrapper, on the wavefront pipeline (k_wf_ext puts the previous success aside,
    k_wf_stale resolves its flag; one batch and several), the megakernel and the closest-hit kernel."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    from tests.scene_helpers import stale_front_face_scene, films_equal
    hs = api.HostScene(stale_front_face_scene(tmp_path, enclosed), str(tmp_path))
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    W = H = 48
    cam = hs.camera(W, H)
    r = np.random.default_rng(5)
    m = 20000
    o = np.tile(np.array([[0.5, 1.5, 8.0]], np.float32), (m, 1))
    d = (np.stack([r.uniform(-0.35, 0.25, m), r.uniform(-0.4, 0.1, m), np.full(m, -1.0)], 1)).astype(np.float32)
    o[m // 2:] = np.array([0.3, 0.5, -2.5], np.float32); d[m // 2:, 2] = 1.0
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        p = api.default_params(8, 8, 1, quirks=q)
        g, c = dev.closest_hit(p, o, d), world.closest_hit(p, o, d)
        for k in ("prim", "tri", "front_face"):
            assert np.array_equal(g[k], c[k]), (k, q)
        assert np.array_equal(g["normal"].view(np.uint32), c["normal"].view(np.uint32))
        ref, sr = world.render_tile(cam, api.default_params(W, H, 4, quirks=q, stats=True))
        for tail, mega, slots in (("1", False, None), ("1000", False, None), ("1000", False, str(W * H)), ("1", True, None)):
            monkeypatch.setenv("HRT_WF_TAIL_ROUND", tail)
            if slots: monkeypatch.setenv("HRT_WF_MAX_SLOTS", slots)
            img, st = dev.render_tile(cam, api.default_params(W, H, 4, quirks=q, stats=True, megakernel=mega))
            if slots: monkeypatch.delenv("HRT_WF_MAX_SLOTS")
            assert (st.rays, st.mesh_hits) == (sr.rays, sr.mesh_hits), (q, tail, mega, slots)
            assert films_equal(img, ref), (q, tail, mega, slots)
    dev.close()


def _lbvh_check(pos, nodes, order, depth, max_leaf):
    """Walks the tree hrt_bvh_build_device returned: every position of the leaf order in exactly one leaf, leaves of at most
    max_leaf triangles, child boxes = the padded ITriangle boxes below, united and widened by host/bvh_build.cpp's rounding guard
    (bit for bit), depth = inner levels."""
    n = pos.shape[0]
    assert sorted(order.tolist()) == list(range(n))
    p = pos[order].reshape(n, 3, 3)
    lo = p.min(1) - np.float32(1e-4); hi = p.max(1) + np.float32(1e-4)
    f = nodes.view(np.float32); ch = nodes.view(np.int32)
    seen = np.zeros(n, dtype=np.int32)
    # [c0_min_x, c0_max_x, c0_min_y, c0_max_y, c1_min_x, c1_max_x, c1_min_y, c1_max_y, c0_min_z, c0_max_z, c1_min_z, c1_max_z, child0, child1]
    cols = {0: ((0, 2, 8), (1, 3, 9)), 1: ((4, 6, 10), (5, 7, 11))}
    visited = np.zeros(len(nodes), dtype=bool)

    def guard(mn, mx):
        g = np.float32(1e-6) + np.float32(4e-7) * np.maximum(np.abs(mn), np.abs(mx))
        return (mn - g).astype(np.float32), (mx + g).astype(np.float32)

    def walk(ref):      # -> (box min, box max, first, last, depth)
        if ref < 0:
            enc = ~ref
            first, cnt = enc >> 3, (enc & 7) + 1
            assert cnt <= max_leaf and first + cnt <= n
            seen[first:first + cnt] += 1
            return lo[first:first + cnt].min(0), hi[first:first + cnt].max(0), first, first + cnt - 1, 0
        assert 0 <= ref < len(nodes) and not visited[ref]
        visited[ref] = True
        out = []
        for c in (0, 1):
            mn, mx, a, b, d = walk(int(ch[ref, 12 + c]))
            gmn, gmx = guard(mn, mx)
            assert np.array_equal(f[ref, list(cols[c][0])].view(np.uint32), gmn.view(np.uint32)), (ref, c)
            assert np.array_equal(f[ref, list(cols[c][1])].view(np.uint32), gmx.view(np.uint32)), (ref, c)
            out.append((mn, mx, a, b, d))
        assert out[0][3] + 1 == out[1][2]                      # the children's ranges are adjacent, left first
        return np.minimum(out[0][0], out[1][0]), np.maximum(out[0][1], out[1][1]), out[0][2], out[1][3], 1 + max(out[0][4], out[1][4])

    import sys
    sys.setrecursionlimit(10000)
    _, _, a, b, d = walk(0)
    assert (a, b) == (0, n - 1) and (seen == 1).all() and visited.all() and d == depth
    return d


@pytest.mark.parametrize("algo", ["lbvh", "sah"])
def test_device_bvh_builder_structure(built, tmp_path, algo):
    """hrt_bvh_build_device (csrc/hrt_lbvh.hip: a Morton-ordered LBVH) and hrt_bvh_build_sah (csrc/hrt_sahbvh.hip: the host's
    binned-SAH algorithm on the device).  Structure, boxes and depth on the teapot, on soups with coincident centroids (equal
    Morton codes: ties are split by position; one centroid for all: median splits) and on the smallest inputs."""
    from hobbyraytracer_amd import api
    api.write_teapot_obj(str(tmp_path / "teapot.obj"), 1.0)
    hs = api.HostScene(_one_mesh_scene(tmp_path, "teapot.obj"), str(tmp_path))
    m, (pos, _, _) = hs.flat.meshes[0], hs.mesh_arrays(0)
    cases = [("teapot", np.asarray(pos, dtype=np.float32).reshape(-1, 9))]
    r = np.random.default_rng(2)
    cases.append(("soup", (r.uniform(-1, 1, (5000, 1, 3)) + r.normal(scale=0.05, size=(5000, 3, 3))).astype(np.float32).reshape(-1, 9)))
    same = np.tile(r.uniform(-1, 1, (1, 9)).astype(np.float32), (300, 1))                      # 300 copies of one triangle
    cases.append(("copies", same))
    cases.append(("copies+1", np.concatenate([same, r.uniform(-5, 5, (1, 9)).astype(np.float32)])))
    for k in (3, 4, 5, 9):
        cases.append((f"n={k}", r.uniform(-1, 1, (k, 9)).astype(np.float32)))
    cases.append(("flat", np.concatenate([r.uniform(-1, 1, (2000, 3, 2)), np.zeros((2000, 3, 1))], 2).astype(np.float32).reshape(-1, 9)))
    for name, tri in cases:
        for max_leaf in (1, 2, 4):
            if tri.shape[0] <= max_leaf: continue
            nodes, order, depth = api.bvh_build_device(tri, max_leaf, algo=algo)
            d = _lbvh_check(tri, nodes, order, depth, max_leaf)
            assert d <= 31, (name, d)
    with pytest.raises(api.HrtError):
        api.bvh_build_device(np.zeros((2, 9), np.float32), 2, algo=algo)      # nothing to build: the host wraps such a mesh in one leaf
    bad = cases[1][1].copy(); bad[7, 3] = np.nan
    with pytest.raises(api.HrtError):
        api.bvh_build_device(bad, 2, algo=algo)


def _one_mesh_scene(tmp_path, obj):
    (tmp_path / "one.yaml").write_text(
        "film:\n    width: 32\n    height: 32\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [0, 2, 9]\n    look_at: [0, 1, 0]\n    up: [0, 1, 0]\n    fov: 40\n    aperture: 0\n    focal_distance: 9\n"
        "    background: [0.6, 0.7, 0.9]\n"
        "materials:\n  - name: a\n    type: lambertian\n    albedo: [0.8, 0.4, 0.3]\n"
        "objects:\n  - type: mesh\n    path: %s\n    material: a\n" % obj)
    return str(tmp_path / "one.yaml")


@pytest.mark.parametrize("algo", ["lbvh", "sah"])
def test_device_bvh_builder_renders_like_the_host_builder(built, assets, scenes_dir, algo):
    """The film does not depend on the culling tree: with a tree from the GPU the fixed-quirks film of the teapot scene is the host
    tree's bit for bit, and with the reference's quirks (whose self-hit winners follow the reference tree restated over the soup in
    LEAF order, which moves with the builder) it is the oracle's on the same flattened scene.  hrt_bvh_build_sah builds THE host
    builder's tree (same node count, same depth, the very same number of box and triangle tests); the LBVH is looser."""
    from hobbyraytracer_amd import api
    from oracle import oracle_py as orc
    W = H = 96
    p_fixed = api.default_params(W, H, 8, quirks=api.QUIRKS_FIXED, stats=True)
    hs0 = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    dev0 = api.DeviceScene(hs0.flat_ptr, 0)
    sah, s_sah = dev0.render_tile(hs0.camera(W, H), p_fixed)
    dev0.close()
    api.use_device_bvh_builder(True, algo=algo)
    try:
        hs = api.HostScene(f"{scenes_dir}/teapot_scene.yaml", assets)
    finally:
        api.use_device_bvh_builder(False)
    if algo == "lbvh":
        assert hs.flat.n_nodes != hs0.flat.n_nodes or hs.bvh_depth(0) != hs0.bvh_depth(0)   # another tree
    else:
        assert hs.flat.n_nodes == hs0.flat.n_nodes and hs.bvh_depth(0) == hs0.bvh_depth(0)   # the same tree
    dev, world = api.DeviceScene(hs.flat_ptr, 0), orc.World(hs.flat_ptr)
    cam = hs.camera(W, H)
    img, st = dev.render_tile(cam, p_fixed)
    assert np.array_equal(img.view(np.uint32), sah.view(np.uint32)) and st.rays == s_sah.rays
    if algo == "lbvh":
        assert st.box_tests > s_sah.box_tests                                                   # ... a looser one
    else:
        assert (st.box_tests, st.tri_tests) == (s_sah.box_tests, s_sah.tri_tests)               # ... the host's, test for test
    for q in (api.QUIRKS_REFERENCE, api.QUIRKS_FIXED):
        ref, sr = world.render_tile(cam, api.default_params(W, H, 8, quirks=q, stats=True))
        for mega in (False, True):
            img, st = dev.render_tile(cam, api.default_params(W, H, 8, quirks=q, stats=True, megakernel=mega))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.rays == sr.rays, (q, mega)
    dev.close()
