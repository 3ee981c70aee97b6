// main.cpp — the CLI, drop-in for the reference's executable (main.cpp:142-195):
//     hobbyraytracer [scene.yaml]
// Default scene "teapot_scene.yaml" in the cwd, same console lines, same exit
// code convention (Q-12: the process returns Film::outputFilm()'s int, 1 on
// success; -1 when the scene fails to load).  Additive flags:
//     --gpus N   --seed S   --spp N   --size WxH   --quirks reference|fixed
//     --assets DIR (search dir for meshes / textures)   --stats   --out FILE
//     --make-assets DIR (write the procedural teapot.obj / marble_bust_01.obj / old_hall_4k.hdr and exit)
//     --progressive N (take the samples in passes of N and rewrite the output image after every pass)
//     --checkpoint FILE (store the accumulation buffer after every pass)   --resume (continue from that file)
//     --max-passes K (stop after K passes; with --checkpoint the render can be resumed later)
//     --dump-linear FILE.pfm (the fp32 linear film, bit for bit, next to the tonemapped image)
//     --obj-indices reference|rebased (multi-object OBJ files: the reference's un-rebased face indices, mesh.cpp:111-114, or correct ones)
//     --lens (thin-lens sampling with the scene's `aperture`: camera.h:34's commented-out circularRand(lensRadius); off = the reference)
//     --no-progress (no reporter thread and no progress counter on the device: main.cpp:97-109), --progress-ms N (its interval, 500)
//     --rccl (gather the film through an RCCL communicator even on one GPU; with --gpus N > 1 it always is)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>

#include "../../include/hrt_host.h"
#include "assets.h"
#include "classes.h"
#include "image_io.h"
#include "render.h"

using namespace hrthost;

constexpr int NUM_THREADS = 12;  // main.cpp:34 (unused by render, as in the reference)

static void printElapsed(const char* what, std::chrono::high_resolution_clock::time_point start) {
    auto eMS = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start);
    unsigned int iH = (unsigned)std::chrono::duration_cast<std::chrono::hours>(eMS).count();
    unsigned int iM = (unsigned)std::chrono::duration_cast<std::chrono::minutes>(eMS).count() - (iH * 60);
    long double fS = (eMS.count() / 1000.0) - (double)((iM * 60) + (iH * 3600));
    std::cout << std::endl << std::setprecision(6) << what << " (completed in " << iH << ":" << iM << ":" << fS << ")" << std::endl;
}

int main(int argc, char** argv) {
    auto start = std::chrono::high_resolution_clock::now();
    std::string file = "teapot_scene.yaml";  // main.cpp:146
    std::string assets, out, makeAssets, dumpLinear;
    RenderOptions opt;
    int spp = -1, sw = -1, sh = -1;
    bool haveFile = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::cerr << flag << " needs a value" << std::endl; std::exit(2); }
            return argv[++i];
        };
        if (a == "--gpus") opt.gpus = std::atoi(next("--gpus"));
        else if (a == "--seed") opt.seed = std::strtoull(next("--seed"), nullptr, 0);
        else if (a == "--spp") spp = std::atoi(next("--spp"));
        else if (a == "--size") { if (std::sscanf(next("--size"), "%dx%d", &sw, &sh) != 2) { std::cerr << "--size WxH" << std::endl; return 2; } }
        else if (a == "--quirks") { std::string q = next("--quirks"); opt.quirks = (q == "fixed") ? HRT_QUIRKS_FIXED : HRT_QUIRKS_REFERENCE; }
        else if (a == "--assets") assets = next("--assets");
        else if (a == "--out") out = next("--out");
        else if (a == "--stats") opt.stats = true;
        else if (a == "--no-progress") opt.progress = false;
        else if (a == "--progress-ms") opt.progress_interval_ms = std::max(1, std::atoi(next("--progress-ms")));
        else if (a == "--rccl") opt.force_rccl = true;
        else if (a == "--lens") opt.thin_lens = true;
        else if (a == "--obj-indices") { std::string v = next("--obj-indices"); setenv("HRT_OBJ_INDICES", v == "rebased" ? "rebased" : "reference", 1); }
        else if (a == "--bvh") {     // who builds the meshes' culling trees: the host (binned SAH, default) or the GPU (gpu-sah: the same tree; lbvh: fastest to build, +16 % box tests)
            const std::string v = next("--bvh");
            if (v == "lbvh") hrt_host_set_bvh_builder(hrt_bvh_build_device, 0);
            else if (v == "gpu-sah") hrt_host_set_bvh_builder(hrt_bvh_build_sah, 0);
            else if (v != "sah") { std::cerr << "--bvh takes sah, gpu-sah or lbvh" << std::endl; return -1; }
        }
        else if (a == "--make-assets") makeAssets = next("--make-assets");
        else if (a == "--progressive") opt.pass_samples = std::atoi(next("--progressive"));
        else if (a == "--checkpoint") opt.checkpoint = next("--checkpoint");
        else if (a == "--resume") opt.resume = true;
        else if (a == "--max-passes") opt.max_passes = std::atoi(next("--max-passes"));
        else if (a == "--dump-linear") dumpLinear = next("--dump-linear");
        else if (!haveFile) { file = a; haveFile = true; }
    }
    if (!makeAssets.empty()) {
        long t = writeTeapotObj(makeAssets + "/teapot.obj", 1.0);
        long b = writeBustObj(makeAssets + "/marble_bust_01.obj", 1.0);
        bool h = writeHallHdr(makeAssets + "/old_hall_4k.hdr", 4096, 2048);
        std::cout << "teapot.obj: " << t << " triangles, marble_bust_01.obj: " << b << " triangles, old_hall_4k.hdr: " << (h ? "ok" : "FAILED") << std::endl;
        return (t > 0 && b > 0 && h) ? 0 : 1;
    }

    Scene scene;
    if (scene.loadScene(file, assets) < 1) return -1;  // main.cpp:155-156

    std::shared_ptr<Film> film = scene.getFilm();
    if (sw > 0 || spp > 0) {
        film_desc f = film->getFilm();
        if (sw > 0 && (sw < 2 || sh < 2 || (long long)sw * sh > (1ll << 30))) { std::cerr << "--size: 2x2 up to 2^30 pixels" << std::endl; return 2; }
        scene.setFilmSize(sw > 0 ? sw : f.width, sh > 0 ? sh : f.height, spp > 0 ? spp : f.samples);
    }
    if (!out.empty()) film->setOutput(out);
    Camera camera = scene.getCamera();
    std::shared_ptr<Texture> background = scene.getBackground();
    std::shared_ptr<HittableList> world = scene.getScene();

    printElapsed(("Loaded scene: " + file + "!").c_str(), start);
    const double load_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count();

    if (opt.resume && opt.checkpoint.empty()) { std::cerr << "--resume needs --checkpoint FILE" << std::endl; return 2; }
    if (opt.pass_samples > 0) opt.on_pass = [&film](int) { film->outputFilm(); };   // preview image after every pass
    hrt_stats stats{};
    double seconds = 0.0;
    hrt_status st = render(NUM_THREADS, background, world, camera, film, opt, &stats, &seconds);
    if (st != HRT_OK) return -1;

    int r = film->outputFilm();
    if (!dumpLinear.empty() && !writePFM(dumpLinear, film->linear().data(), film->getFilm().width, film->getFilm().height)) {
        std::cerr << "cannot write " << dumpLinear << std::endl;
        return -1;
    }

    if (opt.stats || std::getenv("HRT_STATS")) {
        const double bytes = 32.0 * stats.box_tests + 36.0 * stats.tri_tests + 60.0 * stats.mesh_hits + 12.0 * stats.env_lookups +
                             12.0 * film->getFilm().width * film->getFilm().height;
        // wall_s: the reference's own stopwatch (main.cpp:144,184): process start to after the image file is written
        const double wall_s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count();
        std::printf("{\"rays\": %llu, \"samples\": %llu, \"box_tests\": %llu, \"tri_tests\": %llu, \"render_s\": %.6f, "
                    "\"kernel_ms\": %.3f, \"mrays_per_s\": %.3f, \"msamples_per_s\": %.3f, \"algorithmic_gb_per_s\": %.3f, "
                    "\"load_s\": %.6f, \"wall_s\": %.6f, \"gpus\": %d}\n",
                    (unsigned long long)stats.rays, (unsigned long long)stats.samples, (unsigned long long)stats.box_tests,
                    (unsigned long long)stats.tri_tests, seconds, stats.kernel_ms, stats.rays / seconds / 1e6,
                    stats.samples / seconds / 1e6, stats.kernel_ms > 0 ? bytes / (stats.kernel_ms * 1e-3) / 1e9 : 0.0, load_s, wall_s, opt.gpus);
    }
    printElapsed("Done!", start);
    return r;  // main.cpp:194 (1 = success)
}
