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
#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

namespace hrthost {

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
    std::vector<hrt_stats> pstats(G);
    std::vector<hrt_status> pst(G, HRT_OK);
    std::vector<std::string> perr(G);
    auto work = [&](int g) {
        const int rows = hrt_stripe_rows(f.height, R, g, G);
        parts[g].assign((size_t)rows * f.width * 3, 0.0f);
        pst[g] = hrt_render_stripes(scenes[g], &cam, &pr, R, g, G, parts[g].data(), &pstats[g]);
        if (pst[g] != HRT_OK) perr[g] = hrt_last_error();
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < G; ++g) threads.emplace_back(work, g);
    work(0);
    for (auto& t : threads) t.join();
    const auto t1 = std::chrono::high_resolution_clock::now();
    if (render_seconds) *render_seconds = std::chrono::duration<double>(t1 - t0).count();

    hrt_stats total{};
    for (int g = 0; g < G; ++g) {
        if (pst[g] != HRT_OK) {
            std::cerr << "\nrender on device " << g << " failed: " << hrt_status_str(pst[g]) << ": " << perr[g] << std::endl;
            for (hrt_scene* s : scenes) hrt_scene_destroy(s);
            return pst[g];
        }
        total.rays += pstats[g].rays; total.samples += pstats[g].samples; total.box_tests += pstats[g].box_tests;
        total.tri_tests += pstats[g].tri_tests; total.mesh_hits += pstats[g].mesh_hits; total.env_lookups += pstats[g].env_lookups;
        total.launches += pstats[g].launches;
        if (pstats[g].kernel_ms > total.kernel_ms) total.kernel_ms = pstats[g].kernel_ms;
    }
    // gather: rank g's local row l is absolute row hrt_stripe_row_index(...)
    std::vector<float>& lin = film->linear();
    for (int g = 0; g < G; ++g) {
        const int rows = hrt_stripe_rows(f.height, R, g, G);
        for (int l = 0; l < rows; ++l) {
            const int row = hrt_stripe_row_index(f.height, R, g, G, l);
            std::memcpy(&lin[(size_t)row * f.width * 3], &parts[g][(size_t)l * f.width * 3], (size_t)f.width * 3 * sizeof(float));
        }
    }
    st = hrt_resolve_u8(scenes[0], lin.data(), numPixels, film->getPixels());
    if (st != HRT_OK) std::cerr << "\nresolve failed: " << hrt_last_error() << std::endl;
    std::cout << "\rPixels rendered: " << numPixels << "/" << numPixels << std::flush << "\n";
    for (hrt_scene* s : scenes) hrt_scene_destroy(s);
    if (stats) *stats = total;
    return st;
}

}  // namespace hrthost
