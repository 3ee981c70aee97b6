"""Generates tests/golden/stb_written.hdr, stb_written.png and io_expected.npz with the REFERENCE's own
vendored stb (oracle/_ref/libstbref.so, compiled by oracle/Makefile from /root/reference/dependencies/stb
where it lies).  The fixtures are data: files written by stb plus the pixels stb itself decodes from them.
Run in the build container (needs /root/reference):  python tests/golden/make_io_fixtures.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstbref.so"))
    lib.stbi_loadf.restype = C.POINTER(C.c_float)
    lib.stbi_loadf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.stbi_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.stbi_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    r = np.random.default_rng(2024)
    hdr = np.exp(r.uniform(-6, 4, (20, 48, 3))).astype(np.float32)
    hdr[3:6, 5:30] = 12.0
    hdr[15] = 0.0
    png = r.integers(0, 256, (19, 31, 3), dtype=np.uint8)
    png[4:9, :] = (png[4:9, :] // 32) * 32   # smooth-ish rows so stb picks non-zero filters
    hp, pp = os.path.join(HERE, "stb_written.hdr"), os.path.join(HERE, "stb_written.png")
    assert lib.stbi_write_hdr(hp.encode(), 48, 20, 3, hdr.ctypes.data_as(C.POINTER(C.c_float))) == 1
    assert lib.stbi_write_png(pp.encode(), 31, 19, 3, png.ctypes.data, 31 * 3) == 1
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    ptr = lib.stbi_loadf(hp.encode(), C.byref(w), C.byref(h), C.byref(ch), 0)
    dec = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    np.savez(os.path.join(HERE, "io_expected.npz"), hdr_decoded_by_stb=dec, png_pixels=png)
    print("wrote", hp, pp)
