// jpeg_lite.cpp — JPEG decoding for ImageTexture (texture.cpp:30-51 loads through stbi_load, which reads JPEG).
//
// Scope: what image textures come as — 8-bit baseline / extended-sequential and PROGRESSIVE Huffman JPEG, 1 (grey) or 3
// (YCbCr, or RGB when an Adobe marker says so) components, sampling factors 1 or 2 per axis, restart intervals,
// interleaved or per-component scans.  Arithmetic-coded, lossless, 12-bit and CMYK files are refused with a message.
//
// Entropy decoding is the procedure of ITU T.81 Annex F (any conforming decoder yields the same coefficients).  From
// the coefficients on, the PIXELS depend on the decoder, so those steps follow the arithmetic of stb_image v2.27, the
// decoder behind the reference's stbi_load, and tests/test_image_io.py pins the result against it bit for bit:
//   * dequantised coefficients are truncated to 16 bits;
//   * inverse DCT: the "islow" integer algorithm with 12-bit constants, 2 extra bits after the column pass
//     (round at 1 << 9, shift 10), row pass rounded at 1 << 16, level shift 128 folded in, shift 17, clamp;
//   * chroma upsampling: triangle filters centred JFIF-style — vertical (3 near + far + 2) >> 2, horizontal the same
//     with the row ends copied, 2x2 as (3a + b + 8) >> 4 of the vertical sums; other factors repeat samples;
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits and shifted by 8, the Cb term of G
//     masked to its upper 16 bits.
#include "jpeg_lite.h"

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace hrthost {
namespace {

struct HuffTable {
    bool present = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];   // per code length 1..16 (T.81 F.2.2.3)
    void build(const uint8_t counts[16]) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w = 0, hgt = 0;          // samples that matter
    int w2 = 0, h2 = 0;          // allocated (whole MCUs)
    int dc_pred = 0;
    std::vector<uint8_t> data;
    std::vector<int16_t> coeff;  // progressive: all blocks' coefficients (natural order), (w2 / 8) x (h2 / 8) x 64
};

struct Decoder {
    const uint8_t* p; const uint8_t* end;
    std::string err;
    uint16_t dequant[4][64];     // natural order
    bool have_q[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[3];
    int n_comp = 0, width = 0, height = 0, hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    bool jfif = false; int adobe_transform = -1;
    bool progressive = false;
    int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;   // progressive scan parameters (T.81 G.1)
    bool rgb_ids = false;
    // bit reader
    uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false; int marker = -1;

    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
    int u8() { return p < end ? *p++ : -1; }
    int u16() { int a = u8(), b = u8(); return (a < 0 || b < 0) ? -1 : (a << 8) | b; }

    void reset_bits() { bitbuf = 0; bitcnt = 0; hit_marker = false; marker = -1; }
    int bit() {
        if (bitcnt == 0) {
            int b = 0;
            if (!hit_marker) {
                b = u8();
                if (b < 0) { b = 0; hit_marker = true; }
                else if (b == 0xFF) {
                    int c = u8();
                    while (c == 0xFF) c = u8();          // fill bytes
                    if (c != 0) { marker = c; hit_marker = true; b = 0; }
                }
            }
            bitbuf = (uint32_t)b; bitcnt = 8;
        }
        --bitcnt;
        return (bitbuf >> bitcnt) & 1;
    }
    int receive(int n) { int v = 0; for (int i = 0; i < n; ++i) v = (v << 1) | bit(); return v; }
    static int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }
    int decode(const HuffTable& t) {
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | bit();
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) return t.vals[t.valptr[len] + code - t.mincode[len]];
        }
        return -1;
    }
};

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

// One 8-point pass of the integer inverse DCT ("islow"); constants are round(c * 4096).  Inputs s[0..7]; returns the
// even part in x[0..3] and the odd part in t[0..3], both scaled by 4096, to be combined by the caller.
// Corrupt files can hold coefficients far outside what an encoder produces; all sums and products below are taken
// modulo 2^32 (unsigned arithmetic, converted back for the arithmetic shifts), which is what the reference's decoder does
// in practice on such data, instead of overflowing a signed int.
struct W {                                        // int with wrapping +, -, *
    uint32_t u;
    W() : u(0) {}
    W(int x) : u((uint32_t)x) {}
    int i() const { return (int)u; }
    friend W operator+(W a, W b) { W r; r.u = a.u + b.u; return r; }
    friend W operator-(W a, W b) { W r; r.u = a.u - b.u; return r; }
    friend W operator*(W a, W b) { W r; r.u = a.u * b.u; return r; }
    W& operator+=(W b) { u += b.u; return *this; }
};
inline W fx(double c) { return W((int)(c * 4096 + 0.5)); }
inline void idct_1d(const W s[8], W x[4], W t[4]) {
    W p2 = s[2], p3 = s[6];
    W p1 = (p2 + p3) * fx(0.5411961);
    W t2 = p1 + p3 * fx(-1.847759065);
    W t3 = p1 + p2 * fx(0.765366865);
    p2 = s[0]; p3 = s[4];
    W t0 = (p2 + p3) * W(4096), t1 = (p2 - p3) * W(4096);
    x[0] = t0 + t3; x[3] = t0 - t3; x[1] = t1 + t2; x[2] = t1 - t2;
    t0 = s[7]; t1 = s[5]; t2 = s[3]; t3 = s[1];
    p3 = t0 + t2; W p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
    W p5 = (p3 + p4) * fx(1.175875602);
    t0 = t0 * fx(0.298631336); t1 = t1 * fx(2.053119869); t2 = t2 * fx(3.072711026); t3 = t3 * fx(1.501321110);
    p1 = p5 + p1 * fx(-0.899976223); p2 = p5 + p2 * fx(-2.562915447);
    p3 = p3 * fx(-1.961570560); p4 = p4 * fx(-0.390180644);
    t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
}
void idct_block(uint8_t* out, int stride, const int16_t d[64]) {
    W v[64];
    for (int c = 0; c < 8; ++c) {                 // columns: keep 2 extra bits
        W s[8], x[4], t[4];
        for (int r = 0; r < 8; ++r) s[r] = W(d[8 * r + c]);
        idct_1d(s, x, t);
        for (int k = 0; k < 4; ++k) x[k] += W(512);
        v[0 * 8 + c] = W((x[0] + t[3]).i() >> 10); v[7 * 8 + c] = W((x[0] - t[3]).i() >> 10);
        v[1 * 8 + c] = W((x[1] + t[2]).i() >> 10); v[6 * 8 + c] = W((x[1] - t[2]).i() >> 10);
        v[2 * 8 + c] = W((x[2] + t[1]).i() >> 10); v[5 * 8 + c] = W((x[2] - t[1]).i() >> 10);
        v[3 * 8 + c] = W((x[3] + t[0]).i() >> 10); v[4 * 8 + c] = W((x[3] - t[0]).i() >> 10);
    }
    for (int r = 0; r < 8; ++r) {                 // rows: remove 2^17, add the level shift
        W x[4], t[4];
        idct_1d(&v[8 * r], x, t);
        for (int k = 0; k < 4; ++k) x[k] += W(65536 + (128 << 17));
        uint8_t* o = out + (size_t)r * stride;
        o[0] = clamp8((x[0] + t[3]).i() >> 17); o[7] = clamp8((x[0] - t[3]).i() >> 17);
        o[1] = clamp8((x[1] + t[2]).i() >> 17); o[6] = clamp8((x[1] - t[2]).i() >> 17);
        o[2] = clamp8((x[2] + t[1]).i() >> 17); o[5] = clamp8((x[2] - t[1]).i() >> 17);
        o[3] = clamp8((x[3] + t[0]).i() >> 17); o[4] = clamp8((x[3] - t[0]).i() >> 17);
    }
}

bool decode_block(Decoder& z, Component& c, int16_t d[64]) {
    std::memset(d, 0, 64 * sizeof(int16_t));
    const HuffTable& hd = z.dc[c.td]; const HuffTable& ha = z.ac[c.ta];
    const uint16_t* q = z.dequant[c.tq];
    int t = z.decode(hd);
    if (t < 0 || t > 15) return z.fail("bad huffman code");
    const int diff = t ? Decoder::extend(z.receive(t), t) : 0;
    c.dc_pred += diff;
    d[0] = (int16_t)(c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
        const int rs = z.decode(ha);
        if (rs < 0) return z.fail("bad huffman code");
        const int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (rs != 0xF0) break;
            k += 16;
        } else {
            k += r;
            if (k > 63) return z.fail("bad huffman code");
            const int zz = kZigzag[k++];
            d[zz] = (int16_t)(Decoder::extend(z.receive(s), s) * q[zz]);
        }
    }
    return true;
}

// Progressive scans (T.81 Annex G): DC first / refinement, AC first / refinement with end-of-band runs.  Coefficients
// are stored UNquantised-times-2^Al; dequantisation happens once all scans are in (finish_progressive).
bool prog_dc(Decoder& z, Component& c, int16_t* d) {
    if (z.se != 0) return z.fail("DC and AC coefficients in one progressive scan");
    if (z.ah == 0) {
        std::memset(d, 0, 64 * sizeof(int16_t));
        const int t = z.decode(z.dc[c.td]);
        if (t < 0 || t > 15) return z.fail("bad huffman code");
        c.dc_pred += t ? Decoder::extend(z.receive(t), t) : 0;
        d[0] = (int16_t)(c.dc_pred * (1 << z.al));
    } else if (z.bit()) d[0] = (int16_t)(d[0] + (int16_t)(1 << z.al));
    return true;
}
bool prog_ac(Decoder& z, Component& c, int16_t* d) {
    if (z.ss == 0) return z.fail("DC and AC coefficients in one progressive scan");
    const HuffTable& ha = z.ac[c.ta];
    if (z.ah == 0) {
        if (z.eob_run) { --z.eob_run; return true; }
        int k = z.ss;
        do {
            const int rs = z.decode(ha);
            if (rs < 0) return z.fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    z.eob_run = 1 << r;
                    if (r) z.eob_run += z.receive(r);
                    --z.eob_run;
                    break;
                }
                k += 16;
            } else {
                k += r;
                if (k > 63) return z.fail("bad huffman code");
                d[kZigzag[k++]] = (int16_t)(Decoder::extend(z.receive(s), s) * (1 << z.al));
            }
        } while (k <= z.se);
        return true;
    }
    const int16_t bit = (int16_t)(1 << z.al);
    auto refine = [&](int16_t& v) {                   // one correction bit for an already non-zero coefficient
        if (z.bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    };
    if (z.eob_run) {
        --z.eob_run;
        for (int k = z.ss; k <= z.se; ++k) { int16_t& v = d[kZigzag[k]]; if (v != 0) refine(v); }
        return true;
    }
    int k = z.ss;
    do {
        const int rs = z.decode(ha);
        if (rs < 0) return z.fail("bad huffman code");
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
            if (r < 15) {
                z.eob_run = (1 << r) - 1;
                if (r) z.eob_run += z.receive(r);
                r = 64;                                // run to the end of the band, refining on the way
            }                                          // r == 15: sixteen zero coefficients, then go on
        } else {
            if (s != 1) return z.fail("bad huffman code");
            s = z.bit() ? bit : -bit;
        }
        while (k <= z.se) {
            int16_t& v = d[kZigzag[k++]];
            if (v != 0) refine(v);
            else {
                if (r == 0) { v = (int16_t)s; break; }
                --r;
            }
        }
    } while (k <= z.se);
    return true;
}
void finish_progressive(Decoder& z) {
    for (int n = 0; n < z.n_comp; ++n) {
        Component& c = z.comp[n];
        const int bw = (c.w + 7) >> 3, bh = (c.hgt + 7) >> 3, cw = c.w2 / 8;
        for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) {
                int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * cw)];
                for (int k = 0; k < 64; ++k) d[k] = (int16_t)(d[k] * z.dequant[c.tq][k]);
                idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, d);
            }
    }
}

bool decode_scan(Decoder& z, const int* order, int n_scan) {
    z.reset_bits();
    z.eob_run = 0;
    for (int i = 0; i < z.n_comp; ++i) z.comp[i].dc_pred = 0;
    int todo = z.restart_interval ? z.restart_interval : 0x7fffffff;
    int16_t d[64];
    // one block: baseline decodes and transforms it at once; progressive updates its stored coefficients
    auto block = [&](Component& c, int bx, int by) -> bool {
        if (!z.progressive) {
            if (!decode_block(z, c, d)) return false;
            idct_block(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, d);
            return true;
        }
        int16_t* cd = &c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 / 8))];
        return z.ss == 0 ? prog_dc(z, c, cd) : prog_ac(z, c, cd);
    };
    auto restart_point = [&]() -> int {           // 1: continue after RSTn, 0: scan over, -1: error
        if (--todo > 0) return 1;
        // the entropy-coded segment ends at a marker; drop the padding bits
        while (!z.hit_marker) {
            // scan forward for the marker that should follow immediately
            z.bitcnt = 0;
            (void)z.bit();
        }
        if (z.marker < 0xD0 || z.marker > 0xD7) return 0;
        z.reset_bits();
        z.eob_run = 0;
        for (int i = 0; i < z.n_comp; ++i) z.comp[i].dc_pred = 0;
        todo = z.restart_interval ? z.restart_interval : 0x7fffffff;
        return 1;
    };
    if (n_scan == 1) {
        Component& c = z.comp[order[0]];
        const int bw = (c.w + 7) >> 3, bh = (c.hgt + 7) >> 3;
        for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) {
                if (!block(c, i, j)) return false;
                if (z.restart_interval && restart_point() == 0) return true;
            }
        return true;
    }
    for (int j = 0; j < z.mcu_y; ++j)
        for (int i = 0; i < z.mcu_x; ++i) {
            for (int k = 0; k < n_scan; ++k) {
                Component& c = z.comp[order[k]];
                for (int y = 0; y < c.v; ++y)
                    for (int x = 0; x < c.h; ++x) {
                        if (!block(c, i * c.h + x, j * c.v + y)) return false;
                    }
            }
            if (z.restart_interval && restart_point() == 0) return true;
        }
    return true;
}

// ---- chroma upsampling: one output row of `w` low-res samples -> hs * w samples
const uint8_t* resample(std::vector<uint8_t>& buf, const uint8_t* near_, const uint8_t* far_, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near_;
    uint8_t* out = buf.data();
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near_[0]; return out; }
        out[0] = near_[0];
        out[1] = (uint8_t)((near_[0] * 3 + near_[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) {
            const int n = 3 * near_[i] + 2;
            out[i * 2 + 0] = (uint8_t)((n + near_[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + near_[i + 1]) >> 2);
        }
        out[i * 2 + 0] = (uint8_t)((near_[w - 2] * 3 + near_[w - 1] + 2) >> 2);
        out[i * 2 + 1] = near_[w - 1];
    } else if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return out; }
        int t1 = 3 * near_[0] + far_[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near_[i] + far_[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    } else {
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near_[i];
    }
    return out;
}

inline int f2fx(float c) { return ((int)(c * 4096.0f + 0.5f)) << 8; }

}  // namespace

bool decodeJPEG(const uint8_t* bytes, size_t n_bytes, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    Decoder z;
    z.p = bytes; z.end = bytes + n_bytes;
    auto bad = [&](const std::string& m) { err = "JPEG: " + (z.err.empty() ? m : z.err); return false; };
    if (z.u8() != 0xFF || z.u8() != 0xD8) return bad("not a JPEG file");
    bool have_frame = false, done = false;
    while (!done) {
        int m = z.u8();
        if (m < 0) return bad("truncated file");
        if (m != 0xFF) continue;
        while (m == 0xFF) m = z.u8();
        if (m < 0) return bad("truncated file");
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;           // standalone markers
        const int len = z.u16();
        if (len < 2 || z.p + (len - 2) > z.end) return bad("bad marker length");
        const uint8_t* seg_end = z.p + (len - 2);
        switch (m) {
            case 0xDB: {   // DQT
                while (z.p < seg_end) {
                    const int pq = z.u8(), t = pq & 15, sixteen = pq >> 4;
                    if (t > 3 || sixteen > 1) return bad("bad DQT table");
                    for (int i = 0; i < 64; ++i) {
                        const int v = sixteen ? z.u16() : z.u8();
                        if (v < 0) return bad("truncated DQT");
                        z.dequant[t][kZigzag[i]] = (uint16_t)v;
                    }
                    z.have_q[t] = true;
                }
                break;
            }
            case 0xC4: {   // DHT
                while (z.p < seg_end) {
                    const int q = z.u8(), tc = q >> 4, th = q & 15;
                    if (tc > 1 || th > 3) return bad("bad DHT header");
                    uint8_t counts[16];
                    int total = 0;
                    for (int i = 0; i < 16; ++i) { const int c = z.u8(); if (c < 0) return bad("truncated DHT"); counts[i] = (uint8_t)c; total += c; }
                    if (total > 256 || z.p + total > seg_end) return bad("bad DHT table");
                    HuffTable& t = tc ? z.ac[th] : z.dc[th];
                    for (int i = 0; i < total; ++i) t.vals[i] = (uint8_t)z.u8();
                    t.build(counts);
                }
                break;
            }
            case 0xC0: case 0xC1: case 0xC2: {   // SOF0 / SOF1 / SOF2: baseline, extended sequential, progressive (Huffman)
                z.progressive = m == 0xC2;
                if (have_frame) return bad("several frames");
                if (z.u8() != 8) return bad("only 8-bit samples are supported");
                z.height = z.u16(); z.width = z.u16(); z.n_comp = z.u8();
                if (z.height <= 0 || z.width <= 0) return bad("bad image size");
                if (z.n_comp != 1 && z.n_comp != 3) return bad("only grey and 3-component files are supported (got " + std::to_string(z.n_comp) + " components)");
                z.rgb_ids = true;
                for (int i = 0; i < z.n_comp; ++i) {
                    Component& c = z.comp[i];
                    c.id = z.u8();
                    const int hv = z.u8();
                    c.h = hv >> 4; c.v = hv & 15; c.tq = z.u8();
                    if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return bad("bad component header");
                    if (z.n_comp == 3 && c.id != "RGB"[i]) z.rgb_ids = false;
                    z.hmax = c.h > z.hmax ? c.h : z.hmax; z.vmax = c.v > z.vmax ? c.v : z.vmax;
                }
                if (z.n_comp != 3) z.rgb_ids = false;
                for (int i = 0; i < z.n_comp; ++i)
                    if (z.hmax % z.comp[i].h || z.vmax % z.comp[i].v) return bad("unsupported sampling factors");
                z.mcu_x = (z.width + z.hmax * 8 - 1) / (z.hmax * 8);
                z.mcu_y = (z.height + z.vmax * 8 - 1) / (z.vmax * 8);
                {   // every 8x8 block costs at least one bit of entropy-coded data (its DC code; a progressive file can skip
                    // the rest with end-of-band runs): a header that promises more blocks than the file can hold -- with a
                    // factor 2 to spare -- is refused before planes of that size are allocated
                    uint64_t blocks = 0;
                    for (int i = 0; i < z.n_comp; ++i) blocks += (uint64_t)z.mcu_x * z.mcu_y * z.comp[i].h * z.comp[i].v;
                    if (blocks / 16 > (uint64_t)n_bytes) return bad("image size does not fit the file");
                }
                for (int i = 0; i < z.n_comp; ++i) {
                    Component& c = z.comp[i];
                    c.w = (z.width * c.h + z.hmax - 1) / z.hmax;
                    c.hgt = (z.height * c.v + z.vmax - 1) / z.vmax;
                    c.w2 = z.mcu_x * c.h * 8; c.h2 = z.mcu_y * c.v * 8;
                    c.data.assign((size_t)c.w2 * c.h2, 0);
                    if (z.progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);   // (w2 / 8) * (h2 / 8) * 64
                }
                have_frame = true;
                break;
            }
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return bad("lossless / hierarchical / arithmetic-coded JPEG is not supported");
            case 0xDD: z.restart_interval = z.u16(); break;
            case 0xE0: if (len >= 7 && std::memcmp(z.p, "JFIF\0", 5) == 0) z.jfif = true; break;
            case 0xEE: if (len >= 14 && std::memcmp(z.p, "Adobe\0", 6) == 0) z.adobe_transform = z.p[11]; break;
            case 0xDA: {   // SOS + entropy-coded data
                if (!have_frame) return bad("scan before frame header");
                const int ns = z.u8();
                if (ns < 1 || ns > z.n_comp) return bad("bad scan header");
                int order[3];
                for (int i = 0; i < ns; ++i) {
                    const int id = z.u8(), tt = z.u8();
                    int which = -1;
                    for (int k = 0; k < z.n_comp; ++k) if (z.comp[k].id == id) which = k;
                    if (which < 0) return bad("scan names an unknown component");
                    z.comp[which].td = tt >> 4; z.comp[which].ta = tt & 15;
                    if (z.comp[which].td > 3 || z.comp[which].ta > 3 || !z.have_q[z.comp[which].tq]) return bad("scan uses a table the file does not define");
                    order[i] = which;
                }
                z.ss = z.u8(); z.se = z.u8();
                { const int a = z.u8(); z.ah = a >> 4; z.al = a & 15; }
                if (z.progressive) {
                    if (z.ss > 63 || z.se > 63 || z.ss > z.se || z.ah > 13 || z.al > 13) return bad("bad progressive scan parameters");
                    if (z.ss > 0 && ns != 1) return bad("interleaved AC scan");
                } else {
                    z.ss = 0; z.se = 63; z.ah = z.al = 0;
                }
                for (int i = 0; i < ns; ++i) {
                    const Component& c = z.comp[order[i]];
                    const bool need_dc = !z.progressive || (z.ss == 0 && z.ah == 0), need_ac = !z.progressive || z.ss > 0;
                    if ((need_dc && !z.dc[c.td].present) || (need_ac && !z.ac[c.ta].present)) return bad("scan uses a table the file does not define");
                }
                z.p = seg_end;
                if (!decode_scan(z, order, ns)) return bad("corrupt entropy-coded data");
                if (z.hit_marker && z.marker == 0xD9) done = true;
                else if (z.hit_marker && z.marker >= 0) { z.p -= 2; }   // let the loop see the marker again
                continue;
            }
            default: break;   // other APPn, COM, ...
        }
        z.p = seg_end;
    }
    if (!have_frame) return bad("no frame header");
    if (z.progressive) finish_progressive(z);

    // ---- upsample + colour conversion, row by row
    w = z.width; h = z.height;
    rgb.assign((size_t)w * h * 3, 0);
    const bool is_rgb = z.n_comp == 3 && (z.rgb_ids || (z.adobe_transform == 0 && !z.jfif));
    struct Up { int hs, vs, ystep, ypos, wlo; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } up[3];
    for (int k = 0; k < z.n_comp; ++k) {
        Up& r = up[k];
        r.hs = z.hmax / z.comp[k].h; r.vs = z.vmax / z.comp[k].v;
        r.ystep = r.vs >> 1; r.ypos = 0; r.wlo = (w + r.hs - 1) / r.hs;
        r.line0 = r.line1 = z.comp[k].data.data();
        r.buf.assign((size_t)w + 8 + (size_t)r.hs * r.wlo, 0);
    }
    const int c_r = f2fx(1.40200f), c_gr = -f2fx(0.71414f), c_gb = -f2fx(0.34414f), c_b = f2fx(1.77200f);
    for (int j = 0; j < h; ++j) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < z.n_comp; ++k) {
            Up& r = up[k];
            const bool bottom = r.ystep >= (r.vs >> 1);
            row[k] = resample(r.buf, bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.wlo, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z.comp[k].hgt) r.line1 += z.comp[k].w2;
            }
        }
        uint8_t* o = &rgb[(size_t)j * w * 3];
        if (z.n_comp == 1) {
            for (int i = 0; i < w; ++i) { o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = row[0][i]; }
        } else if (is_rgb) {
            for (int i = 0; i < w; ++i) { o[3 * i] = row[0][i]; o[3 * i + 1] = row[1][i]; o[3 * i + 2] = row[2][i]; }
        } else {
            for (int i = 0; i < w; ++i) {
                const int yf = (row[0][i] << 20) + (1 << 19);
                const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                int r = yf + cr * c_r;
                int g = yf + cr * c_gr + (int)((uint32_t)(cb * c_gb) & 0xffff0000u);
                int b = yf + cb * c_b;
                r >>= 20; g >>= 20; b >>= 20;
                o[3 * i] = clamp8(r); o[3 * i + 1] = clamp8(g); o[3 * i + 2] = clamp8(b);
            }
        }
    }
    return true;
}

bool readJPEG(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<uint8_t> bytes;
    uint8_t chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) bytes.insert(bytes.end(), chunk, chunk + n);
    std::fclose(f);
    return decodeJPEG(bytes.data(), bytes.size(), rgb, w, h, err);
}

}  // namespace hrthost
