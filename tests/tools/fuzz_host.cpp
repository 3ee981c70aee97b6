// tests/tools/fuzz_host.cpp -- mutation fuzzer for the host-side file readers (TEST TOOL, not a test and not product code).
//   build:  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude tests/tools/fuzz_host.cpp \
//               hobbyraytracer_amd/host/{classes,bvh_build,scene,yaml_lite,image_io,jpeg_lite,assets,host_api}.cpp -lz -o /tmp/fuzz_host
//   run:    /tmp/fuzz_host <png|jpeg|hdr|yaml|obj> <iterations> <seed> <work dir> <seed file>...
// Every mutated file goes through the same C entry point the product uses (include/hrt_host.h); the readers must return
// an error or a result, never crash, hang, overrun or hit undefined behaviour.  The mutant being read is kept as
// <work dir>/cur.<ext>, so after a sanitizer abort it is still there.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "hrt_host.h"

static std::vector<uint8_t> slurp(const char* p) {
    std::vector<uint8_t> v;
    FILE* f = std::fopen(p, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", p); std::exit(2); }
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    std::fclose(f);
    return v;
}
static void spit(const std::string& p, const std::vector<uint8_t>& v) {
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", p.c_str()); std::exit(2); }
    if (!v.empty()) std::fwrite(v.data(), 1, v.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: fuzz_host kind iterations seed workdir files...\n"); return 2; }
    const std::string kind = argv[1];
    const long iters = std::atol(argv[2]);
    std::mt19937_64 rng(std::strtoull(argv[3], nullptr, 10));
    const std::string dir = argv[4];
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 5; i < argc; ++i) seeds.push_back(slurp(argv[i]));
    const bool text = kind == "yaml" || kind == "obj";
    const std::string ext = kind == "jpeg" ? "jpg" : kind;
    const std::string cur = dir + "/cur." + ext;
    static const char* tokens[] = {"nan", "inf", "-inf", "1e39", "-1e39", "1e-46", "0", "-0", "4294967296", "-2147483649", "[", "]", ":", "-", "  ",
                                   "\n", "\t", "#", "f 1/2/3 4/5/6 7/8/9", "f 0 0 0", "f -1 -2 -3", "f 99999999 1 2", "v", "vn", "vt", "{", "}", "'", "\"",
                                   "transform:", "material:", "type: mesh", "type: sphere", "path: ", "samples: ", "width: ", "0x7fffffff", "1e400", "."};
    std::vector<uint8_t> out(64u << 20);
    long ok = 0, err = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> m = seeds[rng() % seeds.size()];
        const int n_mut = 1 + (int)(rng() % 6);
        for (int k = 0; k < n_mut && !m.empty(); ++k) {
            const size_t pos = rng() % m.size();
            switch (rng() % (text ? 9 : 7)) {
                case 0: m[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: m[pos] = (uint8_t)rng(); break;
                case 2: m[pos] = (uint8_t)((rng() & 1) ? 0xff : 0x00); break;
                case 3: m.resize(pos); break;                                                     // truncate
                case 4: { const size_t len = 1 + rng() % 16; if (pos + len <= m.size()) m.erase(m.begin() + pos, m.begin() + pos + len); break; }
                case 5: { const size_t len = 1 + rng() % 64, src = rng() % m.size();              // splice a copy
                          std::vector<uint8_t> chunk(m.begin() + src, m.begin() + std::min(m.size(), src + len));
                          m.insert(m.begin() + pos, chunk.begin(), chunk.end()); break; }
                case 6: { if (pos + 4 <= m.size()) { const uint32_t v[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x00010000u, 0xffffu};  // length-like fields
                              const uint32_t x = v[rng() % 7]; const bool be = rng() & 1;
                              for (int b = 0; b < 4; ++b) m[pos + b] = (uint8_t)(x >> (be ? 24 - 8 * b : 8 * b)); } break; }
                case 7: { const char* t = tokens[rng() % (sizeof tokens / sizeof *tokens)]; m.insert(m.begin() + pos, t, t + std::strlen(t)); break; }
                case 8: { size_t e = pos; while (e < m.size() && m[e] != '\n') ++e;                // drop / duplicate a line
                          size_t b = pos; while (b > 0 && m[b - 1] != '\n') --b;
                          if (rng() & 1) m.erase(m.begin() + b, m.begin() + std::min(m.size(), e + 1));
                          else { std::vector<uint8_t> line(m.begin() + b, m.begin() + std::min(m.size(), e + 1)); m.insert(m.begin() + b, line.begin(), line.end()); }
                          break; }
            }
        }
        hrt_status st;
        int32_t w = 0, h = 0;
        if (kind == "png") { spit(cur, m); st = hrt_host_read_png(cur.c_str(), &w, &h, out.data(), (int64_t)out.size()); }
        else if (kind == "jpeg") { spit(cur, m); st = hrt_host_read_jpeg(cur.c_str(), &w, &h, out.data(), (int64_t)out.size()); }
        else if (kind == "hdr") { spit(cur, m); st = hrt_host_read_hdr(cur.c_str(), &w, &h, (float*)out.data(), (int64_t)(out.size() / 4)); }
        else if (kind == "yaml") {
            spit(cur, m);
            hrt_host_scene* s = nullptr;
            st = hrt_host_load_yaml(cur.c_str(), dir.c_str(), &s);
            if (st == HRT_OK && s) { (void)hrt_host_flat(s); }
            if (s) hrt_host_free(s);
        } else if (kind == "obj") {   // workdir holds obj.yaml whose one mesh is cur.obj
            spit(cur, m);
            hrt_host_scene* s = nullptr;
            st = hrt_host_load_yaml((dir + "/obj.yaml").c_str(), dir.c_str(), &s);
            if (st == HRT_OK && s) { (void)hrt_host_flat(s); }
            if (s) hrt_host_free(s);
        } else { std::fprintf(stderr, "unknown kind\n"); return 2; }
        if (st == HRT_OK) ++ok; else ++err;
        if (it % 5000 == 4999) { std::fprintf(stderr, "%s: %ld done, %ld accepted, %ld refused\n", kind.c_str(), it + 1, ok, err); }
    }
    std::printf("%s: %ld mutants, %ld accepted, %ld refused, no crash\n", kind.c_str(), iters, ok, err);
    return 0;
}
