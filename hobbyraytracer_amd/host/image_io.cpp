// image_io.cpp — the file codecs the reference takes from stb (texture.cpp:30-51,
// 99-115: stbi_load / stbi_loadf; film.cpp:59-79: stbi_write_png/tga/bmp),
// written from the published format descriptions (Radiance RGBE, PNG/zlib, BMP,
// TGA) — stb itself is third-party source inside the reference tree and is
// not copied.  Observable behaviour kept: LDR textures are forced to 3
// channels; .hdr decodes to 3 fp32 channels with value = mantissa * 2^(e-136)
// (stb_image.h:7053-7078); an LDR file opened as an environment map is
// linearised with gamma 2.2 (stbi_loadf's ldr_to_hdr); JPEG is not supported
// (reported as a load failure, which the reference renders as cyan).
#include "image_io.h"
#include "jpeg_lite.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace hrthost {

namespace {
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    // a directory opens like a file and reports a size of 2^63 - 1 (or -1); nothing this reads is anywhere near a terabyte
    if (!f || n < 0 || (unsigned long long)n > (1ull << 40)) return false;
    out.resize((size_t)n);
    if (n) f.read((char*)out.data(), n);
    return (bool)f;
}
bool ends_with(const std::string& str, const std::string& suffix) {  // hobbyraytracer.h:40-42
    return str.size() >= suffix.size() && str.compare(str.size() - suffix.size(), suffix.size(), suffix) == 0;
}
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
}  // namespace

// ------------------------------------------------------------------ Radiance .hdr
bool readHDR(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> d;
    if (!read_file(path, d)) { err = "cannot open " + path; return false; }
    size_t p = 0;
    auto getline = [&](std::string& s) {
        s.clear();
        while (p < d.size() && d[p] != '\n') s += (char)d[p++];
        if (p < d.size()) ++p;
        return true;
    };
    std::string line;
    getline(line);
    if (line != "#?RADIANCE" && line != "#?RGBE") { err = "not a Radiance HDR file"; return false; }
    bool fmt = false;
    for (;;) {
        if (p >= d.size()) { err = "truncated HDR header"; return false; }
        getline(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) { err = "unsupported HDR format"; return false; }
    getline(line);
    int hh = 0, ww = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0) { err = "unsupported HDR layout"; return false; }
    w = ww; h = hh;
    // stb_image's own limit (STBI_MAX_DIMENSIONS), and a header may not promise more pixels than the bytes that follow can
    // encode (a run-length coded scanline spends at least 4 + 8 bytes per 127 pixels, a flat one 4 per pixel): nothing
    // of a size the file cannot back is allocated
    if (w > (1 << 24) || h > (1 << 24)) { err = "HDR too large"; return false; }
    {
        const uint64_t row_min = (w < 8 || w >= 32768) ? (uint64_t)w * 4 : 4 + 8 * (((uint64_t)w + 126) / 127);
        if ((uint64_t)h * row_min > (uint64_t)(d.size() - p)) { err = "truncated HDR"; return false; }
    }
    rgb.assign((size_t)w * h * 3, 0.0f);
    auto convert = [](const uint8_t* in, float* out) {  // stbi__hdr_convert
        if (in[3] != 0) {
            float f1 = std::ldexp(1.0f, (int)in[3] - (128 + 8));
            out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1;
        } else { out[0] = out[1] = out[2] = 0.0f; }
    };
    std::vector<uint8_t> scan((size_t)w * 4);
    bool flat = (w < 8 || w >= 32768);
    if (!flat) {
        if (p + 4 > d.size()) { err = "truncated HDR"; return false; }
        if (d[p] != 2 || d[p + 1] != 2 || (d[p + 2] & 0x80)) flat = true;
    }
    if (flat) {
        if (p + (size_t)w * h * 4 > d.size()) { err = "truncated HDR"; return false; }
        for (size_t i = 0; i < (size_t)w * h; ++i) convert(&d[p + 4 * i], &rgb[3 * i]);
        return true;
    }
    for (int j = 0; j < h; ++j) {
        if (p + 4 > d.size()) { err = "truncated HDR"; return false; }
        if (d[p] != 2 || d[p + 1] != 2 || (d[p + 2] & 0x80)) { err = "corrupt HDR scanline"; return false; }
        int len = (d[p + 2] << 8) | d[p + 3];
        if (len != w) { err = "corrupt HDR scanline width"; return false; }
        p += 4;
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                if (p >= d.size()) { err = "truncated HDR"; return false; }
                int count = d[p++];
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || i + count > w || p >= d.size()) { err = "corrupt HDR run"; return false; }
                    uint8_t v = d[p++];
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = v;
                } else {
                    if (count == 0 || i + count > w || p + count > d.size()) { err = "corrupt HDR run"; return false; }
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = d[p++];
                }
            }
        }
        for (int i = 0; i < w; ++i) convert(&scan[(size_t)i * 4], &rgb[((size_t)j * w + i) * 3]);
    }
    return true;
}

bool writePFM(const std::string& path, const float* rgb, int w, int h) {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);                       // negative scale = little-endian
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; --y) ok = std::fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    return (std::fclose(f) == 0) && ok;
}
bool readPFM(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    char magic[3] = {0, 0, 0};
    float scale = 0.0f;
    if (std::fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || std::strcmp(magic, "PF") != 0 || w < 1 || h < 1 || !(scale < 0.0f)) {
        std::fclose(f);
        err = path + " is not a little-endian colour PFM";
        return false;
    }
    std::fgetc(f);                                                      // the single whitespace after the scale
    {   // the raster must be there before anything of its size is allocated
        const long at = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        const long end = std::ftell(f);
        std::fseek(f, at, SEEK_SET);
        if (at < 0 || end < at || end == 0x7fffffffffffffffl || (uint64_t)w * (uint64_t)h * 12u > (uint64_t)(end - at)) { std::fclose(f); err = path + " is truncated"; return false; }
    }
    rgb.resize((size_t)w * h * 3);
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; --y) ok = std::fread(&rgb[(size_t)y * w * 3], sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    std::fclose(f);
    if (!ok) err = path + " is truncated";
    return ok;
}

bool writeHDR(const std::string& path, const float* rgb, int w, int h) {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "#?RADIANCE\n# written by hrt-mi355x\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    std::vector<uint8_t> px((size_t)w * 4), out;
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i) {
            const float* c = rgb + ((size_t)j * w + i) * 3;
            float m = std::fmax(c[0], std::fmax(c[1], c[2]));
            uint8_t* q = &px[(size_t)i * 4];
            if (!(m >= 1e-32f)) { q[0] = q[1] = q[2] = q[3] = 0; }
            else {
                int e;
                float n = std::frexp(m, &e) * 256.0f / m;
                q[0] = (uint8_t)(c[0] * n); q[1] = (uint8_t)(c[1] * n); q[2] = (uint8_t)(c[2] * n); q[3] = (uint8_t)(e + 128);
            }
        }
        if (w < 8 || w >= 32768) { std::fwrite(px.data(), 1, px.size(), f); continue; }
        out.clear();
        out.push_back(2); out.push_back(2); out.push_back((uint8_t)(w >> 8)); out.push_back((uint8_t)(w & 255));
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                // run of equal bytes?
                int r = 1;
                while (i + r < w && r < 127 && px[(size_t)(i + r) * 4 + k] == px[(size_t)i * 4 + k]) ++r;
                if (r >= 3) { out.push_back((uint8_t)(128 + r)); out.push_back(px[(size_t)i * 4 + k]); i += r; continue; }
                // literal chunk up to the next run of >= 3
                int s = i, n = 0;
                while (s + n < w && n < 128) {
                    int rr = 1;
                    while (s + n + rr < w && rr < 3 && px[(size_t)(s + n + rr) * 4 + k] == px[(size_t)(s + n) * 4 + k]) ++rr;
                    if (rr >= 3) break;
                    ++n;
                }
                if (n == 0) n = 1;
                out.push_back((uint8_t)n);
                for (int z = 0; z < n; ++z) out.push_back(px[(size_t)(s + z) * 4 + k]);
                i += n;
            }
        }
        std::fwrite(out.data(), 1, out.size(), f);
    }
    bool ok = std::ferror(f) == 0;
    std::fclose(f);
    return ok;
}

// ------------------------------------------------------------------ PNG
bool writePNG(const std::string& path, const uint8_t* rgb, int w, int h, int stride) {
    std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
    for (int j = 0; j < h; ++j) {
        raw[(size_t)j * (1 + (size_t)w * 3)] = 0;  // filter: none
        std::memcpy(&raw[(size_t)j * (1 + (size_t)w * 3) + 1], rgb + (size_t)j * stride, (size_t)w * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        put_be32(out, (uint32_t)data.size());
        size_t s = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        put_be32(out, (uint32_t)crc32(0L, out.data() + s, (uInt)(out.size() - s)));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

bool readPNG(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> d;
    if (!read_file(path, d)) { err = "cannot open " + path; return false; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t p = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    w = h = 0;
    while (p + 12 <= d.size()) {
        uint32_t len = be32(&d[p]);
        if (p + 12 + (size_t)len > d.size()) { err = "truncated PNG"; return false; }
        std::string type((char*)&d[p + 4], 4);
        const uint8_t* body = &d[p + 8];
        if (type == "IHDR") {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        p += 12 + (size_t)len;
    }
    // colour type -> channels; allowed bit depths (PNG spec table 11.1)
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (w <= 0 || h <= 0 || !ch || !depth_ok || interlace > 1) { err = "unsupported PNG (colour type / bit depth / interlace method)"; return false; }
    // stb_image's limits (the reference's decoder): STBI_MAX_DIMENSIONS = 1 << 24 per side, and width * height * 4 must
    // fit an int ("too large")
    if (w > (1 << 24) || h > (1 << 24) || (uint64_t)w * (uint64_t)h > (uint64_t)0x7fffffff / 4) { err = "PNG too large"; return false; }
    if (ctype == 3 && plte.empty()) { err = "palette PNG without PLTE"; return false; }
    // Adam7 passes (or the one pass of a non-interlaced file): origin and spacing of the pass's pixels
    static const int kX0[7] = {0, 4, 0, 2, 0, 1, 0}, kY0[7] = {0, 0, 4, 0, 2, 0, 1}, kDX[7] = {8, 8, 4, 4, 2, 2, 1}, kDY[7] = {8, 8, 8, 4, 4, 2, 2};
    const int n_pass = interlace ? 7 : 1;
    const int bits_pp = ch * depth;
    const int fbpp = bits_pp >= 8 ? bits_pp / 8 : 1;           // filter unit: whole bytes per pixel, at least 1
    size_t total = 0;
    int pw[7], ph[7];
    for (int k = 0; k < n_pass; ++k) {
        pw[k] = interlace ? (w - kX0[k] + kDX[k] - 1) / kDX[k] : w;
        ph[k] = interlace ? (h - kY0[k] + kDY[k] - 1) / kDY[k] : h;
        if (pw[k] > 0 && ph[k] > 0) total += ((size_t)(((size_t)pw[k] * bits_pp + 7) / 8) + 1) * ph[k];
    }
    // deflate expands at most 1032 : 1, so a header that promises more than the IDAT data can hold is refused before
    // anything of that size is allocated
    if (total > idat.size() * 1032 + 1024) { err = "PNG inflate failed"; return false; }
    std::vector<uint8_t> raw(total);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) { err = "PNG inflate failed"; return false; }
    rgb.assign((size_t)w * h * 3, 0);
    // stb_image (the reference's decoder, texture.cpp:34-36) reduces 16-bit samples to their high byte and scales grey
    // samples of 1 / 2 / 4 bits to 0..255; palette indices are looked up; alpha is dropped for the 3-channel request
    const int grey_scale = depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1;
    size_t at = 0;
    std::vector<uint8_t> prev, cur;
    for (int k = 0; k < n_pass; ++k) {
        if (pw[k] <= 0 || ph[k] <= 0) continue;
        const size_t row = ((size_t)pw[k] * bits_pp + 7) / 8;
        prev.assign(row, 0); cur.assign(row, 0);
        for (int j = 0; j < ph[k]; ++j) {
            const uint8_t* src = &raw[at]; at += row + 1;
            const int ft = src[0];
            if (ft > 4) { err = "bad PNG filter"; return false; }
            for (size_t i = 0; i < row; ++i) {
                const int a = i >= (size_t)fbpp ? cur[i - fbpp] : 0, b = prev[i], c = i >= (size_t)fbpp ? prev[i - fbpp] : 0;
                const int x = src[1 + i];
                int v;
                switch (ft) {
                    case 0: v = x; break;
                    case 1: v = x + a; break;
                    case 2: v = x + b; break;
                    case 3: v = x + ((a + b) >> 1); break;
                    default: { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                               v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); break; }
                }
                cur[i] = (uint8_t)v;
            }
            const int y = interlace ? kY0[k] + j * kDY[k] : j;
            for (int i = 0; i < pw[k]; ++i) {
                const int x = interlace ? kX0[k] + i * kDX[k] : i;
                uint8_t* o = &rgb[((size_t)y * w + x) * 3];
                auto sample = [&](int c) -> int {           // channel c of pixel i of this scanline, reduced to 8 bits
                    if (depth == 8) return cur[(size_t)i * ch + c];
                    if (depth == 16) return cur[((size_t)i * ch + c) * 2];
                    const int bit = i * depth;                // 1, 2, 4 bits: one channel only
                    return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                };
                if (ctype == 3) {
                    const size_t idx = (size_t)sample(0) * 3;
                    if (idx + 2 < plte.size()) { o[0] = plte[idx]; o[1] = plte[idx + 1]; o[2] = plte[idx + 2]; }
                } else if (ctype == 0 || ctype == 4) {
                    const int g = depth < 8 ? sample(0) * grey_scale : sample(0);
                    o[0] = o[1] = o[2] = (uint8_t)g;
                } else { o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2); }
            }
            prev.swap(cur);
        }
    }
    return true;
}

// ------------------------------------------------------------------ BMP / TGA (24-bit, uncompressed)
bool writeBMP(const std::string& path, const uint8_t* rgb, int w, int h) {
    const int pad = (-(w * 3)) & 3;
    const uint32_t img = (uint32_t)(w * 3 + pad) * h, size = 54 + img;
    std::vector<uint8_t> o(size, 0);
    auto le32 = [&](size_t at, uint32_t v) { o[at] = v; o[at + 1] = v >> 8; o[at + 2] = v >> 16; o[at + 3] = v >> 24; };
    o[0] = 'B'; o[1] = 'M'; le32(2, size); le32(10, 54); le32(14, 40); le32(18, (uint32_t)w); le32(22, (uint32_t)h);
    o[26] = 1; o[28] = 24; le32(34, img);
    for (int j = 0; j < h; ++j) {
        uint8_t* row = &o[54 + (size_t)(h - 1 - j) * (w * 3 + pad)];
        for (int i = 0; i < w; ++i) { const uint8_t* s = rgb + ((size_t)j * w + i) * 3; row[3 * i] = s[2]; row[3 * i + 1] = s[1]; row[3 * i + 2] = s[0]; }
    }
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    return ok;
}
bool writeTGA(const std::string& path, const uint8_t* rgb, int w, int h) {
    std::vector<uint8_t> o(18 + (size_t)w * h * 3, 0);
    o[2] = 2; o[12] = w & 255; o[13] = w >> 8; o[14] = h & 255; o[15] = h >> 8; o[16] = 24; o[17] = 0x20;  // top-left origin
    for (size_t i = 0; i < (size_t)w * h; ++i) { o[18 + 3 * i] = rgb[3 * i + 2]; o[18 + 3 * i + 1] = rgb[3 * i + 1]; o[18 + 3 * i + 2] = rgb[3 * i]; }
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    return ok;
}

// ------------------------------------------------------------------ stbi_load / stbi_loadf stand-ins
bool loadImageRGB8(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::string lower = path;
    for (char& c : lower) c = (char)std::tolower((unsigned char)c);
    if (ends_with(lower, ".png")) return readPNG(path, rgb, w, h, err);
    if (ends_with(lower, ".jpg") || ends_with(lower, ".jpeg")) return readJPEG(path, rgb, w, h, err);
    if (ends_with(lower, ".hdr")) {  // stbi_load on an HDR file: hdr_to_ldr (gamma 1/2.2, scale 1)
        std::vector<float> f;
        if (!readHDR(path, f, w, h, err)) return false;
        rgb.resize(f.size());
        for (size_t i = 0; i < f.size(); ++i) {
            float z = std::pow(f[i], 1.0f / 2.2f) * 255.0f + 0.5f;
            z = z < 0 ? 0 : (z > 255 ? 255 : z);
            rgb[i] = (uint8_t)(int)z;
        }
        return true;
    }
    err = "unsupported image format (.png, .jpg and .hdr are decoded): " + path;
    return false;
}
bool loadImageF32(const std::string& path, std::vector<float>& data, int& w, int& h, int& channels, std::string& err) {
    std::string lower = path;
    for (char& c : lower) c = (char)std::tolower((unsigned char)c);
    if (ends_with(lower, ".hdr")) { channels = 3; return readHDR(path, data, w, h, err); }
    const bool jpg = ends_with(lower, ".jpg") || ends_with(lower, ".jpeg");
    if (ends_with(lower, ".png") || jpg) {  // stbi_loadf on an LDR file: ldr_to_hdr, pow(v/255, 2.2)
        std::vector<uint8_t> rgb;
        if (!(jpg ? readJPEG(path, rgb, w, h, err) : readPNG(path, rgb, w, h, err))) return false;
        channels = 3;
        data.resize(rgb.size());
        for (size_t i = 0; i < rgb.size(); ++i) data[i] = std::pow(rgb[i] / 255.0f, 2.2f);
        return true;
    }
    err = "unsupported image format (.hdr, .png and .jpg are decoded): " + path;
    return false;
}

}  // namespace hrthost
