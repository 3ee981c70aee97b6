// host_api.cpp — C ABI of libhrt_host.so (include/hrt_host.h).
#include <cstring>
#include <iostream>
#include <new>
#include <string>

#include "../../include/hrt_host.h"
#include "assets.h"
#include "classes.h"
#include "image_io.h"
#include "jpeg_lite.h"

using namespace hrthost;

struct hrt_host_scene {
    Scene scene;
    FlatBuilder fb;
    hrt_flat_scene flat;
    std::vector<int> mesh_depth;
};

namespace {
thread_local std::string g_host_err;
hrt_status hfail(hrt_status s, const std::string& m) { g_host_err = m; return s; }
// No C++ exception may cross the C ABI: a reader that runs out of memory on a hostile header (std::bad_alloc,
// std::length_error) reports it like any other unreadable file.
template <class F>
hrt_status guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return hfail(HRT_ERR_OOM, "out of host memory"); }
    catch (const std::exception& e) { return hfail(HRT_ERR_INVALID, e.what()); }
    catch (...) { return hfail(HRT_ERR_INVALID, "unknown C++ exception"); }
}

// collects BVH depths in flatten order by walking the world again
void collectDepths(const hrt_flat_scene& f, std::vector<int>& out) {
    out.assign(f.n_meshes, 0);
    for (uint32_t m = 0; m < f.n_meshes; ++m) {
        const hrt_mesh& mm = f.meshes[m];
        if (!mm.node_count) continue;
        std::vector<std::pair<int32_t, int>> st{{0, 1}};
        int best = 0;
        while (!st.empty()) {
            auto [ni, d] = st.back(); st.pop_back();
            if (d > best) best = d;
            const hrt_bvh_node& n = f.nodes[mm.node_first + ni];
            if (!(n.c0_min_x > n.c0_max_x) && n.child0 >= 0) st.push_back({n.child0, d + 1});
            if (!(n.c1_min_x > n.c1_max_x) && n.child1 >= 0) st.push_back({n.child1, d + 1});
        }
        out[m] = best;
    }
}
}  // namespace

extern "C" {

const char* hrt_host_last_error(void) { return g_host_err.c_str(); }

hrt_status hrt_host_load_yaml(const char* yaml_path, const char* asset_dir, hrt_host_scene** out) {
    if (!yaml_path || !out) return hfail(HRT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    hrt_host_scene* h = new hrt_host_scene;
    try {
        if (h->scene.loadScene(yaml_path, asset_dir ? asset_dir : "") < 1) {
            std::string e = h->scene.lastError;
            delete h;
            return hfail(HRT_ERR_PARSE, e.empty() ? "scene load failed" : e);
        }
        flattenWorld(h->fb, h->scene.getScene(), h->scene.getBackground());
        h->flat = h->fb.flat();
        collectDepths(h->flat, h->mesh_depth);
    } catch (const FlattenError& e) {
        delete h;
        return hfail(e.status, e.what());
    } catch (const std::bad_alloc&) {
        delete h;
        return hfail(HRT_ERR_OOM, "out of host memory");
    } catch (const std::exception& e) {
        delete h;
        return hfail(HRT_ERR_INVALID, e.what());
    } catch (...) {
        delete h;
        return hfail(HRT_ERR_INVALID, "unknown C++ exception");
    }
    *out = h;
    return HRT_OK;
}
void hrt_host_free(hrt_host_scene* s) { delete s; }
const hrt_flat_scene* hrt_host_flat(const hrt_host_scene* s) { return s ? &s->flat : nullptr; }

hrt_status hrt_host_film(const hrt_host_scene* s, int32_t* w, int32_t* h, int32_t* samples, char* output, int32_t cap) {
    return guarded([&]() -> hrt_status {
        if (!s) return hfail(HRT_ERR_INVALID, "NULL scene");
        film_desc f = s->scene.getFilm()->getFilm();
        if (w) *w = f.width;
        if (h) *h = f.height;
        if (samples) *samples = f.samples;
        if (output && cap > 0) { std::strncpy(output, s->scene.getFilm()->output().c_str(), (size_t)cap - 1); output[cap - 1] = 0; }
        return HRT_OK;
    });
}
hrt_status hrt_host_camera(const hrt_host_scene* s, int32_t width, int32_t height, hrt_camera* out) {
    return guarded([&]() -> hrt_status {
        if (!s || !out || width < 1 || height < 1) return hfail(HRT_ERR_INVALID, "bad argument");
        // Scene::setFilmSize mutates; work on the const scene through a copy of the descriptor
        hrt_host_scene* ms = const_cast<hrt_host_scene*>(s);
        film_desc f = ms->scene.getFilm()->getFilm();
        ms->scene.setFilmSize(width, height, f.samples);
        *out = ms->scene.getCamera().flatten();
        ms->scene.setFilmSize(f.width, f.height, f.samples);
        return HRT_OK;
    });
}
void hrt_host_set_bvh_builder(hrt_host_bvh_build_fn fn, int device) { hrthost::setDeviceBvhBuilder((void*)fn, device); }
int32_t hrt_host_bvh_depth(const hrt_host_scene* s, int32_t mesh) {
    if (!s || mesh < 0 || (size_t)mesh >= s->mesh_depth.size()) return -1;
    return s->mesh_depth[mesh];
}

void hrt_default_params(hrt_params* p, int32_t width, int32_t height, int32_t samples) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height; p->samples = samples;
    p->max_depth = 50;       // main.cpp:32
    p->t_min = 0.001f;       // main.cpp:45
    p->quirks = HRT_QUIRKS_REFERENCE;
    p->seed_lo = 0; p->seed_hi = 0; p->flags = 0;
}

int64_t hrt_asset_write_teapot_obj(const char* path, double detail) { try { return path ? writeTeapotObj(path, detail) : -1; } catch (...) { return -1; } }
int64_t hrt_asset_write_bust_obj(const char* path, double detail) { try { return path ? writeBustObj(path, detail) : -1; } catch (...) { return -1; } }
hrt_status hrt_asset_write_hall_hdr(const char* path, int32_t w, int32_t h) {
    return guarded([&]() -> hrt_status {
        if (!path || w < 1 || h < 1) return hfail(HRT_ERR_INVALID, "bad argument");
        return writeHallHdr(path, w, h) ? HRT_OK : hfail(HRT_ERR_IO, std::string("cannot write ") + path);
    });
}

hrt_status hrt_host_write_image(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    return guarded([&]() -> hrt_status {
        if (!path || !rgb || w < 1 || h < 1) return hfail(HRT_ERR_INVALID, "bad argument");
        Film f(w, h, 1, path);
        std::memcpy(f.getPixels(), rgb, (size_t)w * h * 3);
        return f.outputFilm() == 1 ? HRT_OK : hfail(HRT_ERR_IO, std::string("cannot write ") + path);
    });
}
hrt_status hrt_host_read_hdr(const char* path, int32_t* w, int32_t* h, float* out, int64_t cap) {
    return guarded([&]() -> hrt_status {
        if (!path || !w || !h) return hfail(HRT_ERR_INVALID, "bad argument");
        std::vector<float> d; int ww, hh; std::string err;
        if (!readHDR(path, d, ww, hh, err)) return hfail(HRT_ERR_IO, err);
        *w = ww; *h = hh;
        if (out) {
            if (cap < (int64_t)d.size()) return hfail(HRT_ERR_INVALID, "output buffer too small");
            std::memcpy(out, d.data(), d.size() * sizeof(float));
        }
        return HRT_OK;
    });
}
hrt_status hrt_host_read_png(const char* path, int32_t* w, int32_t* h, uint8_t* out, int64_t cap) {
    return guarded([&]() -> hrt_status {
        if (!path || !w || !h) return hfail(HRT_ERR_INVALID, "bad argument");
        std::vector<uint8_t> d; int ww, hh; std::string err;
        if (!readPNG(path, d, ww, hh, err)) return hfail(HRT_ERR_IO, err);
        *w = ww; *h = hh;
        if (out) {
            if (cap < (int64_t)d.size()) return hfail(HRT_ERR_INVALID, "output buffer too small");
            std::memcpy(out, d.data(), d.size());
        }
        return HRT_OK;
    });
}
hrt_status hrt_host_write_hdr(const char* path, const float* rgb, int32_t w, int32_t h) {
    return guarded([&]() -> hrt_status {
        if (!path || !rgb || w < 1 || h < 1) return hfail(HRT_ERR_INVALID, "bad argument");
        return writeHDR(path, rgb, w, h) ? HRT_OK : hfail(HRT_ERR_IO, std::string("cannot write ") + path);
    });
}

hrt_status hrt_host_read_jpeg(const char* path, int32_t* w, int32_t* h, uint8_t* out, int64_t cap) {
    return guarded([&]() -> hrt_status {
        if (!path || !w || !h) return hfail(HRT_ERR_INVALID, "bad argument");
        std::vector<uint8_t> d;
        int ww = 0, hh = 0;
        std::string err;
        if (!readJPEG(path, d, ww, hh, err)) return hfail(HRT_ERR_IO, err);
        *w = ww; *h = hh;
        if (!out) return HRT_OK;
        if (cap < (int64_t)d.size()) return hfail(HRT_ERR_INVALID, "output buffer too small");
        std::memcpy(out, d.data(), d.size());
        return HRT_OK;
    });
}
hrt_status hrt_host_write_pfm(const char* path, const float* rgb, int32_t w, int32_t h) {
    return guarded([&]() -> hrt_status {
        if (!path || !rgb || w < 1 || h < 1) return hfail(HRT_ERR_INVALID, "bad argument");
        return writePFM(path, rgb, w, h) ? HRT_OK : hfail(HRT_ERR_IO, std::string("cannot write ") + path);
    });
}
hrt_status hrt_host_read_pfm(const char* path, int32_t* w, int32_t* h, float* out, int64_t cap) {
    return guarded([&]() -> hrt_status {
        if (!path || !w || !h) return hfail(HRT_ERR_INVALID, "bad argument");
        std::vector<float> rgb;
        int iw = 0, ih = 0;
        std::string err;
        if (!readPFM(path, rgb, iw, ih, err)) return hfail(HRT_ERR_IO, err);
        *w = iw; *h = ih;
        if (!out) return HRT_OK;
        if (cap < (int64_t)rgb.size()) return hfail(HRT_ERR_INVALID, "output buffer too small");
        std::memcpy(out, rgb.data(), rgb.size() * sizeof(float));
        return HRT_OK;
    });
}

}  // extern "C"
