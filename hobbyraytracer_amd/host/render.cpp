// render.cpp — render() (main.cpp:81-140) on the GPU path.
//
// One host thread per GPU (the scene is replicated on every device, SURVEY.md
// §8e); image rows are dealt to devices in interleaved blocks of
// `rows_per_block` rows so that sky rows and geometry rows are shared out
// evenly; the RNG is keyed by the absolute pixel index, so the result is
// bit-identical for every device count.  Rank 0's device resolves the gathered
// linear film to u8 (Film::tonemap + writeColour).
#include "render.h"

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

namespace hrthost {

namespace {
// Checkpoint of a progressive render: header + width*height*3 fp32 sums in absolute row order.
struct Checkpoint {
    int32_t width, height, samples, next_sample;
    uint64_t seed;
    uint32_t quirks;
    int32_t max_depth;
};
const char kMagic[8] = {'H', 'R', 'T', 'C', 'K', 'P', 'T', '1'};

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
    pr.flags = opt.stats ? HRT_FLAG_STATS : 0;

    std::vector<hrt_scene*> scenes(G, nullptr);
    for (int g = 0; g < G; ++g) {
        st = hrt_scene_create(&flat, g, &scenes[g]);
        if (st != HRT_OK) {
            std::cerr << "hrt_scene_create(device " << g << "): " << hrt_status_str(st) << ": " << hrt_last_error() << std::endl;
            for (hrt_scene* s : scenes) hrt_scene_destroy(s);
            return st;
        }
    }

    std::cout << "\rPixels rendered: 0/" << numPixels << std::flush;  // main.cpp:100
    const auto t0 = std::chrono::high_resolution_clock::now();

    const int R = opt.rows_per_block;
    std::vector<std::vector<float>> parts(G);
    for (int g = 0; g < G; ++g) parts[g].assign((size_t)hrt_stripe_rows(f.height, R, g, G) * f.width * 3, 0.0f);
    std::vector<float>& lin = film->linear();     // what the film shows: the preview mean, at the end the final mean
    std::vector<float> sums(lin.size(), 0.0f);     // whole-film accumulation buffer (absolute row order)
    auto gather = [&]() {   // rank g's local row l is absolute row hrt_stripe_row_index(...)
        for (int g = 0; g < G; ++g) {
            const int rows = hrt_stripe_rows(f.height, R, g, G);
            for (int l = 0; l < rows; ++l) {
                const int row = hrt_stripe_row_index(f.height, R, g, G, l);
                std::memcpy(&sums[(size_t)row * f.width * 3], &parts[g][(size_t)l * f.width * 3], (size_t)f.width * 3 * sizeof(float));
            }
        }
    };
    auto scatter = [&]() {
        for (int g = 0; g < G; ++g) {
            const int rows = hrt_stripe_rows(f.height, R, g, G);
            for (int l = 0; l < rows; ++l) {
                const int row = hrt_stripe_row_index(f.height, R, g, G, l);
                std::memcpy(&parts[g][(size_t)l * f.width * 3], &sums[(size_t)row * f.width * 3], (size_t)f.width * 3 * sizeof(float));
            }
        }
    };
    auto cleanup = [&]() { for (hrt_scene* s : scenes) hrt_scene_destroy(s); };

    int s_done = 0;
    if (opt.resume) {
        Checkpoint ck;
        std::string why;
        if (!readCheckpoint(opt.checkpoint, ck, sums, why)) { std::cerr << "\nresume: " << why << std::endl; cleanup(); return HRT_ERR_IO; }
        if (ck.width != f.width || ck.height != f.height || ck.samples != f.samples || ck.seed != opt.seed || ck.quirks != opt.quirks ||
            ck.max_depth != opt.max_depth || ck.next_sample < 0 || ck.next_sample > f.samples) {
            std::cerr << "\nresume: " << opt.checkpoint << " belongs to a different render (film, samples, seed, quirks or depth differ)" << std::endl;
            cleanup();
            return HRT_ERR_INVALID;
        }
        s_done = ck.next_sample;
        scatter();
        std::cout << "\rResumed at sample " << s_done << "/" << f.samples << std::endl;
    }
    const int pass = opt.pass_samples > 0 ? opt.pass_samples : f.samples;
    hrt_stats total{};
    int passes = 0;
    while (s_done < f.samples && (opt.max_passes <= 0 || passes < opt.max_passes)) {
        ++passes;
        const int n = std::min(pass, f.samples - s_done);
        std::vector<hrt_stats> pstats(G);
        std::vector<hrt_status> pst(G, HRT_OK);
        std::vector<std::string> perr(G);
        auto work = [&](int g) {
            pst[g] = hrt_render_stripes_accumulate(scenes[g], &cam, &pr, R, g, G, parts[g].data(), s_done, n, &pstats[g]);
            if (pst[g] != HRT_OK) perr[g] = hrt_last_error();
        };
        std::vector<std::thread> threads;
        for (int g = 1; g < G; ++g) threads.emplace_back(work, g);
        work(0);
        for (auto& t : threads) t.join();
        double pass_kernel_ms = 0.0;
        for (int g = 0; g < G; ++g) {
            if (pst[g] != HRT_OK) {
                std::cerr << "\nrender on device " << g << " failed: " << hrt_status_str(pst[g]) << ": " << perr[g] << std::endl;
                cleanup();
                return pst[g];
            }
            total.rays += pstats[g].rays; total.samples += pstats[g].samples; total.box_tests += pstats[g].box_tests;
            total.tri_tests += pstats[g].tri_tests; total.mesh_hits += pstats[g].mesh_hits; total.env_lookups += pstats[g].env_lookups;
            total.launches += pstats[g].launches; total.traversal_box_tests += pstats[g].traversal_box_tests;
            total.traversal_tri_tests += pstats[g].traversal_tri_tests;
            if (pstats[g].kernel_ms > pass_kernel_ms) pass_kernel_ms = pstats[g].kernel_ms;
        }
        total.kernel_ms += pass_kernel_ms;
        s_done += n;
        gather();
        if (!opt.checkpoint.empty()) {
            Checkpoint ck{f.width, f.height, f.samples, s_done, opt.seed, opt.quirks, opt.max_depth};
            if (!writeCheckpoint(opt.checkpoint, ck, sums)) { std::cerr << "\ncannot write checkpoint " << opt.checkpoint << std::endl; cleanup(); return HRT_ERR_IO; }
        }
        if (s_done < f.samples) {   // preview: mean of the samples so far
            const float k = static_cast<float>(s_done);
            for (size_t i = 0; i < lin.size(); ++i) lin[i] = sums[i] / k;
            st = hrt_resolve_u8(scenes[0], lin.data(), numPixels, film->getPixels());
            if (st != HRT_OK) { std::cerr << "\nresolve failed: " << hrt_last_error() << std::endl; cleanup(); return st; }
            std::cout << "\rSamples rendered: " << s_done << "/" << f.samples << std::flush;
            if (opt.on_pass) opt.on_pass(s_done);
        }
    }
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (render_seconds) *render_seconds = std::chrono::duration<double>(t1 - t0).count();
    if (s_done >= f.samples) lin = sums;           // the last pass divided (main.cpp:126): the sums are the means now
    else if (passes == 0) {                        // nothing rendered in this call (resume of a stopped render with --max-passes 0 ...)
        const float k = static_cast<float>(s_done > 0 ? s_done : 1);
        for (size_t i = 0; i < lin.size(); ++i) lin[i] = sums[i] / k;
    }
    st = hrt_resolve_u8(scenes[0], lin.data(), numPixels, film->getPixels());
    if (st != HRT_OK) std::cerr << "\nresolve failed: " << hrt_last_error() << std::endl;
    std::cout << "\rPixels rendered: " << numPixels << "/" << numPixels << std::flush << "\n";
    for (hrt_scene* s : scenes) hrt_scene_destroy(s);
    if (stats) *stats = total;
    return st;
}

}  // namespace hrthost
