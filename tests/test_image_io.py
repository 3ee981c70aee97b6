"""Codecs behind Texture loading and Film::outputFilm (hobbyraytracer_amd/host/image_io.cpp), pinned
against the reference's own vendored stb (dependencies/stb, compiled from where it lies into
oracle/_ref/libstbref.so by oracle/Makefile; SURVEY.md §8c "partial oracle that does build") and against
golden bytes committed under tests/golden/ (generated with that stb by tests/golden/make_io_fixtures.py),
so the GPU box — which has no /root/reference — still checks them."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _stb():
    p = os.path.join(ROOT, "oracle", "_ref", "libstbref.so")
    if not os.path.exists(p):
        return None
    lib = C.CDLL(p)
    lib.stbi_loadf.restype = C.POINTER(C.c_float)
    lib.stbi_loadf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.stbi_load.restype = C.POINTER(C.c_uint8)
    lib.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.stbi_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.stbi_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    return lib


def _test_image(h=24, w=40, seed=0):
    r = np.random.default_rng(seed)
    img = np.exp(r.uniform(-6, 4, (h, w, 3))).astype(np.float32)
    img[2:5, 3:20] = 7.5       # runs, so the RLE encoder is exercised
    img[10, :] = 0.0
    return img


def test_hdr_roundtrip_own_codec(built, tmp_path):
    from hobbyraytracer_amd import api
    for w in (5, 40, 300):     # flat (< 8 wide) and RLE scanlines
        img = _test_image(13, w, w)
        p = str(tmp_path / f"a{w}.hdr")
        api.write_hdr(p, img)
        back = api.read_hdr(p)
        assert back.shape == img.shape
        m = img.max(2, keepdims=True)
        assert (np.abs(back - img) <= m / 128 + 1e-30).all()     # 8-bit mantissa shared exponent
        p2 = str(tmp_path / f"b{w}.hdr")
        api.write_hdr(p2, back)
        assert np.array_equal(api.read_hdr(p2), back)             # RGBE values are a fixed point


def test_hdr_reader_matches_reference_stb(built, tmp_path):
    """stbi_loadf (texture.cpp:101) on files written by our writer AND by stb's writer."""
    stb = _stb()
    if stb is None:
        pytest.skip("oracle/_ref/libstbref.so not built (reference tree absent): covered by the golden file test")
    from hobbyraytracer_amd import api
    img = _test_image()
    ours, theirs = str(tmp_path / "ours.hdr"), str(tmp_path / "theirs.hdr")
    api.write_hdr(ours, img)
    assert stb.stbi_write_hdr(theirs.encode(), img.shape[1], img.shape[0], 3, img.ctypes.data_as(C.POINTER(C.c_float))) == 1
    for path in (ours, theirs):
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        ptr = stb.stbi_loadf(path.encode(), C.byref(w), C.byref(h), C.byref(ch), 0)
        assert ptr and ch.value == 3
        ref = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
        assert np.array_equal(api.read_hdr(path).view(np.uint32), ref.view(np.uint32)), path


def test_png_writer_decodes_identically_with_reference_stb(built, tmp_path):
    """Film::outputFilm's PNG (film.cpp:63): our file decoded by stb == pixels; stb's file decoded by us == pixels."""
    from hobbyraytracer_amd import api
    r = np.random.default_rng(4)
    img = r.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    ours = str(tmp_path / "ours.png")
    api.write_image(ours, img)
    assert np.array_equal(api.read_png(ours), img)
    stb = _stb()
    if stb is None:
        pytest.skip("oracle/_ref/libstbref.so not built")
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    ptr = stb.stbi_load(ours.encode(), C.byref(w), C.byref(h), C.byref(ch), 3)
    assert ptr and (w.value, h.value) == (53, 37)
    assert np.array_equal(np.ctypeslib.as_array(ptr, shape=(37, 53, 3)), img)
    theirs = str(tmp_path / "theirs.png")
    assert stb.stbi_write_png(theirs.encode(), 53, 37, 3, img.ctypes.data, 53 * 3) == 1
    assert np.array_equal(api.read_png(theirs), img)             # stb uses filters 0-4 per row: exercises the unfilter code


def test_golden_files_written_by_reference_stb(built):
    """Committed fixtures produced by the reference's stb (tests/golden/make_io_fixtures.py)."""
    from hobbyraytracer_amd import api
    exp = np.load(os.path.join(GOLD, "io_expected.npz"))
    assert np.array_equal(api.read_hdr(os.path.join(GOLD, "stb_written.hdr")).view(np.uint32), exp["hdr_decoded_by_stb"].view(np.uint32))
    assert np.array_equal(api.read_png(os.path.join(GOLD, "stb_written.png")), exp["png_pixels"])


def test_bmp_and_tga_writers(built, tmp_path):
    from hobbyraytracer_amd import api
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    api.write_image(str(tmp_path / "a.bmp"), img)
    b = open(tmp_path / "a.bmp", "rb").read()
    assert b[:2] == b"BM" and int.from_bytes(b[18:22], "little") == 7 and int.from_bytes(b[22:26], "little") == 5
    row = 7 * 3 + (-(7 * 3)) % 4
    last = b[54:54 + 21]          # BMP stores bottom row first, BGR
    assert list(last[:3]) == [int(img[4, 0, 2]), int(img[4, 0, 1]), int(img[4, 0, 0])] and len(b) == 54 + row * 5
    api.write_image(str(tmp_path / "a.tga"), img)
    t = open(tmp_path / "a.tga", "rb").read()
    assert t[2] == 2 and t[12] == 7 and t[14] == 5 and list(t[18:21]) == [int(img[0, 0, 2]), int(img[0, 0, 1]), int(img[0, 0, 0])]
    # unknown suffix -> bitmap (film.cpp:73-78)
    api.write_image(str(tmp_path / "a.xyz"), img)
    assert open(tmp_path / "a.xyz", "rb").read()[:2] == b"BM"


def test_pfm_film_dump_is_bit_exact(built, tmp_path):
    """hrt_host_write_pfm / read_pfm: the fp32 film survives bit for bit (NaN payloads, denormals, -0, inf included), rows
    stored bottom first with a little-endian scale line as the format asks."""
    from hobbyraytracer_amd import api
    r = np.random.default_rng(4)
    a = r.standard_normal((7, 5, 3)).astype(np.float32) * np.float32(1e3)
    a[0, 0] = [np.nan, np.inf, -np.inf]
    a[1, 1] = [np.float32(-0.0), np.float32(1e-45), np.float32(3.4e38)]
    a.view(np.uint32)[2, 2, 0] = 0x7fc12345                      # a NaN with a payload
    api.write_pfm(str(tmp_path / "f.pfm"), a)
    raw = open(tmp_path / "f.pfm", "rb").read()
    assert raw.startswith(b"PF\n5 7\n-1.0\n") and len(raw) == len(b"PF\n5 7\n-1.0\n") + a.nbytes
    assert np.array_equal(np.frombuffer(raw[-a.nbytes:], "<f4").reshape(7, 5, 3)[::-1].view(np.uint32), a.view(np.uint32))
    b = api.read_pfm(str(tmp_path / "f.pfm"))
    assert b.shape == a.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    (tmp_path / "bad.pfm").write_bytes(b"Pf\n5 7\n-1.0\n" + bytes(10))
    with pytest.raises(api.HrtError):
        api.read_pfm(str(tmp_path / "bad.pfm"))


def test_jpeg_decoder_matches_the_references_stb(built):
    """hobbyraytracer_amd/host/jpeg_lite.cpp against stbi_load(path, ..., 3) of the stb_image.h the reference vendors
    (texture.cpp:34-36), on files written by the reference's stb_image_write and by libjpeg: 4:4:4, 4:2:2, 4:2:0 (odd sizes,
    one pixel wide, restart intervals, optimised Huffman tables, quality 5 with clamping everywhere), grey, progressive (colour
    4:2:0, 4:4:4 with restarts, grey).  Expected pixels were produced by the reference's decoder itself
    (tests/golden/make_jpeg_fixtures.py).  Bit-exact."""
    from hobbyraytracer_amd import api
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(here, "jpeg_expected.npz"))
    assert len(exp.files) == 13
    for name in exp.files:
        got = api.read_jpeg(os.path.join(here, "jpeg", name))
        assert got.shape == exp[name].shape and np.array_equal(got, exp[name]), name
    with pytest.raises(api.HrtError):
        api.read_jpeg(os.path.join(here, "stb_written.png"))                      # not a JPEG
    raw = open(os.path.join(here, "jpeg", "pil_420_restart.jpg"), "rb").read()
    import tempfile
    d = tempfile.mkdtemp()
    for cut in (3, 40, len(raw) // 2):                                              # truncated files fail cleanly or decode what is there
        open(os.path.join(d, "t.jpg"), "wb").write(raw[:cut])
        try:
            api.read_jpeg(os.path.join(d, "t.jpg"))
        except api.HrtError:
            pass


def test_jpeg_image_texture_in_a_scene(built, tmp_path):
    """ImageTexture with a .jpg path (texture.cpp:30-51 goes through stbi_load, which reads JPEG): the texels are the decoder's."""
    import shutil
    from hobbyraytracer_amd import api
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    shutil.copy(os.path.join(here, "jpeg", "pil_422.jpg"), tmp_path / "tex.jpg")
    (tmp_path / "s.yaml").write_text("""
film:
    width: 8
    height: 8
    samples: 1
    output: o.png
camera:
    position: [0, 0, 3]
    look_at: [0, 0, 0]
    up: [0, 1, 0]
    fov: 50
    aperture: 0
    focal_distance: 3
    background: [0.4, 0.5, 0.6]
textures:
  - name: photo
    type: image
    path: tex.jpg
materials:
  - name: m
    type: lambertian
    albedo: photo
objects:
  - type: sphere
    center: [0, 0, 0]
    radius: 1
    material: m
""")
    hs = api.HostScene(str(tmp_path / "s.yaml"), str(tmp_path))
    f = hs.flat
    tex = [f.textures[i] for i in range(f.n_textures) if f.textures[i].kind == api.TEX_IMAGE][0]
    assert (tex.width, tex.height) == (64, 48)
    texels = np.ctypeslib.as_array(f.texels_u8, shape=(f.n_texels_u8,))[tex.offset:tex.offset + 64 * 48 * 3].reshape(48, 64, 3)
    assert np.array_equal(texels, np.load(os.path.join(here, "jpeg_expected.npz"))["pil_422.jpg"])


def test_png_reader_matches_the_references_stb_on_every_flavour(built):
    """readPNG against stbi_load(path, ..., 3) of the reference's stb on 20 files: grey 1/2/4/8/16 bit, grey+alpha, RGB and
    RGBA at 8 and 16 bit, palettes of 1/2/4/8 bit (one with tRNS), and Adam7-interlaced files down to 1x1 with all five
    filter types (tests/golden/make_png_fixtures.py; expected pixels come from the reference's decoder).  Bit-exact."""
    from hobbyraytracer_amd import api
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(here, "png_expected.npz"))
    assert len(exp.files) == 20
    for name in exp.files:
        got = api.read_png(os.path.join(here, "png", name))
        assert got.shape == exp[name].shape and np.array_equal(got, exp[name]), name


def test_readers_refuse_a_directory(built, tmp_path):
    """A directory opens like a file on Linux and reports a size of 2^63 - 1: the readers must say "cannot open", not try to
    allocate that (found by tests/tools/fuzz_host.cpp: a mutated YAML path pointed the environment-map loader at the
    asset directory itself)."""
    from hobbyraytracer_amd import api
    (tmp_path / "adir").mkdir()
    for fn in (api.read_hdr, api.read_png, api.read_jpeg, api.read_pfm):
        with pytest.raises(api.HrtError):
            fn(str(tmp_path / "adir"))
