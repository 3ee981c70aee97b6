// render.h — the seam of the reference (main.cpp:81-82): same call shape, the
// body is the MI355X path (flatten -> libhrt_hip.so) instead of the PSTL loop.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

#include "classes.h"

namespace hrthost {

struct RenderOptions {
    int gpus = 1;                       // image row blocks are interleaved over this many devices
    int rows_per_block = 8;
    bool thin_lens = false;             // sample the lens as camera.h:34's commented-out circularRand(lensRadius) would (--lens)
    bool force_rccl = false;            // gather through an RCCL communicator even with one device (--rccl; tests)
    uint32_t quirks = HRT_QUIRKS_REFERENCE;
    uint64_t seed = 0;
    int max_depth = 50;                 // MAX_DEPTH (main.cpp:32)
    bool stats = false;                 // count box / triangle tests too
    bool progress = true;               // the reporter thread of main.cpp:97-109: "Pixels rendered: x/N" every 500 ms (--no-progress)
    int progress_interval_ms = 500;     // main.cpp:107
    // Progressive rendering: samples are taken in passes of `pass_samples` (0 = all at once);
    // after every pass but the last the film holds the preview (mean of the samples so far) and `on_pass`
    // is called (the CLI rewrites the output image there).  With `checkpoint` set, the accumulation sums and
    // the next sample index are stored after every pass; `resume` continues a render from such a file (any
    // GPU count: the file holds whole-film rows).  The finished film is bit-identical however it was batched.
    int pass_samples = 0;
    int max_passes = 0;                 // > 0: stop after this many passes (the checkpoint continues the render later)
    std::string checkpoint;
    bool resume = false;
    std::function<void(int samples_done)> on_pass;
};

// render() of main.cpp:81-140.  nThreads is accepted and unused, exactly as in
// the reference (main.cpp:81 never reads it).  Returns HRT_OK or the failing
// status (message on stderr); fills film->getPixels() and film->linear().
hrt_status render(int nThreads, const std::shared_ptr<Texture> background, const std::shared_ptr<Hittable> world,
                  const Camera& camera, std::shared_ptr<Film>& film, const RenderOptions& opt, hrt_stats* stats,
                  double* render_seconds);

}  // namespace hrthost
