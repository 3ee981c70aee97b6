"""Generates tests/golden/png/*.png and png_expected.npz: PNG files of the flavours image textures come in -- grey 1/2/4/8/16
bit, grey+alpha 8/16, RGB 8/16, RGBA 8/16, palette 1/2/4/8 bit (one with tRNS), written by Pillow, and Adam7-INTERLACED
versions of several of them assembled here by hand (zlib + per-pass scanlines, filters 0..4 cycling) -- together with the
pixels the REFERENCE's decoder, stbi_load(path, &w, &h, &n, 3) of the vendored stb_image.h (oracle/_ref/libstbref.so), returns
for them.  Fixtures are data.  Run in the build container (needs /root/reference and Pillow):
    python tests/golden/make_png_fixtures.py"""
import ctypes as C
import os
import struct
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)


def filt(ft, cur, prev, bpp):
    out = bytearray(len(cur))
    for i in range(len(cur)):
        a = cur[i - bpp] if i >= bpp else 0
        b = prev[i]
        c = prev[i - bpp] if i >= bpp else 0
        if ft == 0: p = 0
        elif ft == 1: p = a
        elif ft == 2: p = b
        elif ft == 3: p = (a + b) >> 1
        else:
            pp = a + b - c
            pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
            p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (cur[i] - p) & 255
    return bytes(out)


def pack_row(samples, depth):
    """samples: 1-D array of integer samples of one scanline (all channels interleaved) -> bytes"""
    if depth == 8:
        return bytes(samples.astype(np.uint8))
    if depth == 16:
        return samples.astype(">u2").tobytes()
    bits = "".join(format(int(v), f"0{depth}b") for v in samples)
    bits += "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def write_interlaced(path, img, ctype, depth, plte=None):
    """img: (h, w, ch) integer samples"""
    h, w, ch = img.shape
    x0, y0, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    bpp = max(1, ch * depth // 8)
    raw = b""
    ft = 0
    for k in range(7):
        sub = img[y0[k]::dy[k], x0[k]::dx[k]]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = bytes(len(pack_row(sub[0].reshape(-1), depth)))
        for row in sub:
            cur = pack_row(row.reshape(-1), depth)
            raw += bytes([ft]) + filt(ft, cur, prev, bpp)
            prev = cur
            ft = (ft + 1) % 5
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1))
    if plte is not None:
        data += chunk(b"PLTE", bytes(plte))
    data += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    open(path, "wb").write(data)


if __name__ == "__main__":
    out = os.path.join(HERE, "png")
    os.makedirs(out, exist_ok=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstbref.so"))
    lib.stbi_load.restype = C.POINTER(C.c_ubyte)
    lib.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    r = np.random.default_rng(99)
    H, W = 13, 19
    rgb = r.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rgb[3:7] = (rgb[3:7] // 64) * 64
    a8 = r.integers(0, 256, (H, W), dtype=np.uint8)
    rgb16 = r.integers(0, 65536, (H, W, 3), dtype=np.uint16)
    names = []

    def pil(name, im, **kw):
        im.save(os.path.join(out, name), **kw); names.append(name)
    pil("rgb8.png", Image.fromarray(rgb))
    pil("rgba8.png", Image.fromarray(np.dstack([rgb, a8])))
    pil("grey8.png", Image.fromarray(rgb[:, :, 0]))
    pil("greyalpha8.png", Image.fromarray(np.dstack([rgb[:, :, 0], a8]), "LA"))
    pil("grey16.png", Image.fromarray(rgb16[:, :, 0]))
    pil("grey1.png", Image.fromarray(rgb[:, :, 0] > 127))
    pil("pal8.png", Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=200))
    pil("pal4.png", Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=16), bits=4)
    pil("pal2.png", Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=4), bits=2)
    pil("pal1.png", Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=2), bits=1)
    pil("pal8_trns.png", Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=32), transparency=3)
    # hand-made: 16-bit colour types Pillow does not write, 2/4-bit grey, and Adam7 versions
    pal = r.integers(0, 256, (16, 3), dtype=np.uint8)
    for name, img, ctype, depth, plte in (
            ("i_rgb8.png", rgb.astype(int), 2, 8, None),
            ("i_rgba16.png", np.dstack([rgb16, r.integers(0, 65536, (H, W))]).astype(int), 6, 16, None),
            ("i_rgb16.png", rgb16.astype(int), 2, 16, None),
            ("i_grey2.png", (rgb[:, :, :1] >> 6).astype(int), 0, 2, None),
            ("i_grey4.png", (rgb[:, :, :1] >> 4).astype(int), 0, 4, None),
            ("i_greyalpha16.png", np.dstack([rgb16[:, :, 0], rgb16[:, :, 1]]).astype(int), 4, 16, None),
            ("i_pal4.png", (rgb[:, :, :1] >> 4).astype(int), 3, 4, pal.reshape(-1)),
            ("i_tiny_3x2.png", rgb[:2, :3].astype(int), 2, 8, None),
            ("i_tiny_1x1.png", rgb[:1, :1].astype(int), 2, 8, None)):
        write_interlaced(os.path.join(out, name), img, ctype, depth, plte)
        names.append(name)
    exp = {}
    for name in names:
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        p = lib.stbi_load(os.path.join(out, name).encode(), C.byref(w), C.byref(h), C.byref(n), 3)
        assert p, name
        exp[name] = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    np.savez_compressed(os.path.join(HERE, "png_expected.npz"), **exp)
    print("wrote", len(exp), "fixtures", sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)), "bytes")
