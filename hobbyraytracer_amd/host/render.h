// render.h — the seam of the reference (main.cpp:81-82): same call shape, the
// body is the MI355X path (flatten -> libhrt_hip.so) instead of the PSTL loop.
#pragma once
#include <cstdint>
#include <memory>

#include "classes.h"

namespace hrthost {

struct RenderOptions {
    int gpus = 1;                       // image row blocks are interleaved over this many devices
    int rows_per_block = 8;
    uint32_t quirks = HRT_QUIRKS_REFERENCE;
    uint64_t seed = 0;
    int max_depth = 50;                 // MAX_DEPTH (main.cpp:32)
    bool stats = false;                 // count box / triangle tests too
};

// render() of main.cpp:81-140.  nThreads is accepted and unused, exactly as in
// the reference (main.cpp:81 never reads it).  Returns HRT_OK or the failing
// status (message on stderr); fills film->getPixels() and film->linear().
hrt_status render(int nThreads, const std::shared_ptr<Texture> background, const std::shared_ptr<Hittable> world,
                  const Camera& camera, std::shared_ptr<Film>& film, const RenderOptions& opt, hrt_stats* stats,
                  double* render_seconds);

}  // namespace hrthost
