// render.cpp — render() (main.cpp:81-140) on the GPU path.
//
// The scene is replicated on every device; image rows are dealt
// to devices in interleaved blocks of `rows_per_block` rows so that sky rows
// and geometry rows are shared out evenly; the RNG is keyed by the absolute
// pixel index, so the result is bit-identical for every device count.  The
// devices keep their stripes in their own memory; libhrt_hip.so's multi-GPU
// session (hrt_multi_*) gathers them on the first device over RCCL / xGMI,
// which also resolves the film to u8 (Film::tonemap + writeColour).
#include "render.h"

#include <atomic>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>

namespace hrthost {

namespace {
// Checkpoint of a progressive render: header + width*height*3 fp32 sums in absolute row order.
struct Checkpoint {
    int32_t width, height, samples, next_sample;
    uint64_t seed;
    uint32_t quirks;
    int32_t max_depth;
    uint64_t scene_hash;   // of the flattened scene and the camera: sums of another scene must not be continued
};
const char kMagic[8] = {'H', 'R', 'T', 'C', 'K', 'P', 'T', '2'};

// FNV-1a over everything the kernels read of the scene (hrt_flat_scene's arrays) and the camera constants.
uint64_t sceneHash(const hrt_flat_scene& f, const hrt_camera& cam) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    mix(&cam, sizeof(cam));
    mix(f.prims, sizeof(hrt_prim) * f.n_prims); mix(f.materials, sizeof(hrt_material) * f.n_materials);
    mix(f.textures, sizeof(hrt_texture) * f.n_textures); mix(f.meshes, sizeof(hrt_mesh) * f.n_meshes);
    mix(f.tri_pos, sizeof(float) * 9 * f.n_tris); mix(f.tri_nrm, sizeof(float) * 9 * f.n_tris); mix(f.tri_uv, sizeof(float) * 6 * f.n_tris);
    if (f.tri_ref_order) mix(f.tri_ref_order, sizeof(uint32_t) * f.n_tris);
    mix(f.texels_u8, (size_t)f.n_texels_u8);
    // (a 4096 x 2048 fp32 environment map is 100 MB: sample it -- every 61st float and the size)
    mix(&f.n_texels_f32, sizeof(f.n_texels_f32));
    for (uint64_t i = 0; i < f.n_texels_f32; i += 61) mix(f.texels_f32 + i, sizeof(float));
    mix(&f.background_tex, sizeof(f.background_tex));
    return h;
}

bool writeCheckpoint(const std::string& path, const Checkpoint& ck, const std::vector<float>& sums) {
    const std::string tmp = path + ".tmp";   // never leave a torn file under the real name
    FILE* fp = std::fopen(tmp.c_str(), "wb");
    if (!fp) return false;
    bool ok = std::fwrite(kMagic, 1, 8, fp) == 8 && std::fwrite(&ck, sizeof(ck), 1, fp) == 1 &&
              std::fwrite(sums.data(), sizeof(float), sums.size(), fp) == sums.size();
    ok = (std::fclose(fp) == 0) && ok;
    if (ok) ok = std::rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok;
}
bool readCheckpoint(const std::string& path, Checkpoint& ck, std::vector<float>& sums, std::string& why) {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) { why = "cannot open " + path; return false; }
    char magic[8];
    bool ok = std::fread(magic, 1, 8, fp) == 8 && std::memcmp(magic, kMagic, 8) == 0 && std::fread(&ck, sizeof(ck), 1, fp) == 1;
    if (ok && (ck.width < 2 || ck.height < 2 || (size_t)ck.width * ck.height * 3 != sums.size())) ok = false;
    if (ok) ok = std::fread(sums.data(), sizeof(float), sums.size(), fp) == sums.size();
    std::fclose(fp);
    if (!ok) why = path + " is not a checkpoint of this film";
    return ok;
}
}  // namespace

hrt_status render(int /*nThreads*/, const std::shared_ptr<Texture> background, const std::shared_ptr<Hittable> world,
                  const Camera& camera, std::shared_ptr<Film>& film, const RenderOptions& opt, hrt_stats* stats,
                  double* render_seconds) {
    const film_desc f = film->getFilm();
    const int numPixels = f.width * f.height;

    FlatBuilder fb;
    try {
        flattenWorld(fb, world, background);
    } catch (const FlattenError& e) {
        std::cerr << "flatten: " << e.what() << std::endl;
        return e.status;
    }
    const hrt_flat_scene flat = fb.flat();
    const hrt_camera cam = camera.flatten();

    int ndev = 0;
    hrt_status st = hrt_device_count(&ndev);
    if (st != HRT_OK || ndev < 1) {
        std::cerr << "no MI355X device available: " << hrt_last_error() << " (there is no CPU fallback)" << std::endl;
        return st != HRT_OK ? st : HRT_ERR_NO_DEVICE;
    }
    const int G = opt.gpus < 1 ? 1 : (opt.gpus > ndev ? ndev : opt.gpus);

    hrt_params pr{};
    pr.width = f.width; pr.height = f.height; pr.samples = f.samples;
    pr.max_depth = opt.max_depth; pr.t_min = 0.001f; pr.quirks = opt.quirks;
    pr.seed_lo = (uint32_t)opt.seed; pr.seed_hi = (uint32_t)(opt.seed >> 32);
    pr.flags = (opt.stats ? HRT_FLAG_STATS : 0) | (opt.thin_lens ? HRT_FLAG_THIN_LENS : 0) | (opt.progress ? HRT_FLAG_PROGRESS : 0);

    // The multi-GPU session: scene on every device, stripes accumulated in device memory, RCCL gather (hrt.h hrt_multi_*).
    hrt_multi* multi = nullptr;
    st = hrt_multi_create(&flat, G, nullptr, opt.force_rccl ? 1 : 0, &multi);
    if (st != HRT_OK) {
        std::cerr << "hrt_multi_create(" << G << " devices): " << hrt_status_str(st) << ": " << hrt_last_error() << std::endl;
        return st;
    }

    std::cout << "\rPixels rendered: 0/" << numPixels << std::flush;  // main.cpp:100
    const auto t0 = std::chrono::high_resolution_clock::now();

    // The reporter thread (main.cpp:97-109).  The reference counts finished pixels; here all pixels finish together, so the
    // figure is the pixel-equivalent of the camera paths that have ended: numPixels x paths ended / all paths, read from the
    // devices' host-mapped progress counters (hrt_multi_progress: no HIP call) every 500 ms.  It is woken when the render is
    // over instead of sleeping its interval out (the reference's join can cost its run up to 500 ms).
    std::atomic<long long> pathsBefore{0};          // paths of the passes already finished
    std::mutex rm;
    std::condition_variable rcv;
    bool reporterStop = false;
    std::thread reporter;
    if (opt.progress) {
        reporter = std::thread([&]() {
            const double allPaths = (double)numPixels * (double)f.samples;
            std::unique_lock<std::mutex> lk(rm);
            while (!reporterStop) {
                if (rcv.wait_for(lk, std::chrono::milliseconds(opt.progress_interval_ms), [&] { return reporterStop; })) break;
                uint64_t done = 0, total = 0;
                if (hrt_multi_progress(multi, &done, &total) != HRT_OK) continue;
                if (done > total) done = total;
                long long px = (long long)((double)numPixels * ((double)pathsBefore.load() + (double)done) / allPaths);
                if (px > numPixels) px = numPixels;
                std::cout << "\rPixels rendered: " << px << "/" << numPixels << std::flush;
            }
        });
    }
    auto stopReporter = [&]() {
        if (!reporter.joinable()) return;
        { std::lock_guard<std::mutex> g(rm); reporterStop = true; }
        rcv.notify_all();
        reporter.join();
    };

    const int R = opt.rows_per_block;
    std::vector<float>& lin = film->linear();     // what the film shows: the preview mean, at the end the final mean
    std::vector<float> sums(lin.size(), 0.0f);     // whole-film accumulation buffer (film order), as gathered on the first device
    auto cleanup = [&]() { stopReporter(); hrt_multi_destroy(multi); };

    int s_done = 0;
    bool have_resume = false;
    if (opt.resume) {
        Checkpoint ck;
        std::string why;
        if (!readCheckpoint(opt.checkpoint, ck, sums, why)) { std::cerr << "\nresume: " << why << std::endl; cleanup(); return HRT_ERR_IO; }
        if (ck.width != f.width || ck.height != f.height || ck.samples != f.samples || ck.seed != opt.seed || ck.quirks != opt.quirks ||
            ck.max_depth != opt.max_depth || ck.next_sample < 0 || ck.next_sample > f.samples || ck.scene_hash != sceneHash(flat, cam)) {
            std::cerr << "\nresume: " << opt.checkpoint << " belongs to a different render (scene, camera, film, samples, seed, quirks or depth differ)" << std::endl;
            cleanup();
            return HRT_ERR_INVALID;
        }
        s_done = ck.next_sample;
        pathsBefore.store((long long)numPixels * (long long)s_done);
        have_resume = true;
        std::cout << "\rResumed at sample " << s_done << "/" << f.samples << std::endl;
    }
    const int pass = opt.pass_samples > 0 ? opt.pass_samples : f.samples;
    hrt_stats total{};
    int passes = 0;
    bool resolved = false;
    while (s_done < f.samples && (opt.max_passes <= 0 || passes < opt.max_passes)) {
        ++passes;
        const int n = std::min(pass, f.samples - s_done);
        hrt_stats ps{};
        st = hrt_multi_render(multi, &cam, &pr, R, s_done, n, have_resume ? sums.data() : nullptr, sums.data(), film->getPixels(), &ps);
        have_resume = false;
        if (st != HRT_OK) {
            std::cerr << "\nrender failed: " << hrt_status_str(st) << ": " << hrt_last_error() << std::endl;
            cleanup();
            return st;
        }
        resolved = true;
        total.rays += ps.rays; total.samples += ps.samples; total.box_tests += ps.box_tests; total.tri_tests += ps.tri_tests;
        total.mesh_hits += ps.mesh_hits; total.env_lookups += ps.env_lookups; total.launches += ps.launches;
        total.traversal_box_tests += ps.traversal_box_tests; total.traversal_tri_tests += ps.traversal_tri_tests;
        total.kernel_ms += ps.kernel_ms;
        s_done += n;
        pathsBefore.store((long long)numPixels * (long long)s_done);
        if (!opt.checkpoint.empty()) {
            Checkpoint ck{f.width, f.height, f.samples, s_done, opt.seed, opt.quirks, opt.max_depth, sceneHash(flat, cam)};
            if (!writeCheckpoint(opt.checkpoint, ck, sums)) { std::cerr << "\ncannot write checkpoint " << opt.checkpoint << std::endl; cleanup(); return HRT_ERR_IO; }
        }
        if (s_done < f.samples) {   // preview: mean of the samples so far (the u8 film was resolved on the device)
            const float k = static_cast<float>(s_done);
            for (size_t i = 0; i < lin.size(); ++i) lin[i] = sums[i] / k;
            std::cout << "\rSamples rendered: " << s_done << "/" << f.samples << std::flush;
            if (opt.on_pass) opt.on_pass(s_done);
        }
    }
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (render_seconds) *render_seconds = std::chrono::duration<double>(t1 - t0).count();
    st = HRT_OK;
    if (s_done >= f.samples && resolved) lin = sums;   // the last pass divided (main.cpp:126): the sums are the means now
    else if (!resolved) {                              // nothing rendered in this call (resume of a stopped render with --max-passes 0 ...)
        hrt_stats none{};
        st = hrt_multi_render(multi, &cam, &pr, R, s_done, 0, have_resume ? sums.data() : nullptr, nullptr, film->getPixels(), &none);
        if (st != HRT_OK) std::cerr << "\nresolve failed: " << hrt_last_error() << std::endl;
        const float k = static_cast<float>(s_done > 0 && s_done < f.samples ? s_done : 1);
        for (size_t i = 0; i < lin.size(); ++i) lin[i] = sums[i] / k;
    }
    stopReporter();
    std::cout << "\rPixels rendered: " << numPixels << "/" << numPixels << std::flush << "\n";
    hrt_multi_destroy(multi);
    if (stats) *stats = total;
    return st;
}

}  // namespace hrthost
