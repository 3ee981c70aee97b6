// hrt_hip.hip — HIP kernels for gfx950 (MI355X) and the C ABI of include/hrt.h.
//
// Kernels:
//   k_pathtrace<STATS>  render() + rayColour() of main.cpp:38-140 as a
//                       persistent-lanes kernel: every lane owns one pixel at a
//                       time and walks its samples in order (so the fp32 sum
//                       per pixel has the reference's order, main.cpp:118-126);
//                       when a pixel is finished the lane pulls the next one
//                       from a global counter (one aggregated atomic per wave,
//                       64-bit ballot), so lanes stay busy whatever the path
//                       lengths of their neighbours are.
//   k_resolve           Film::tonemap + Film::writeColour (film.cpp:25-52).
//   k_closest_hit       world->hit() for test rays (parity tests).
//   k_math_probe        the shared math kernels, for CPU==GPU bit tests.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is loaded on demand (rccl_api below)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hrt_device.h"
#include "hrt_pack.h"

using namespace hrt;

// ===================================================================== kernels
namespace {

struct RenderMap {
    int32_t mode;            // 0 = rect tile, 1 = interleaved row blocks
    int32_t x0, y0;          // rect origin
    int32_t rw, rh;          // local region size
    int32_t R, rank, G;      // stripes
    int32_t tiles_x;         // ceil(rw / 8)
    int32_t total_items;     // tiles_x * ceil(rh / 8) * 64
    // exact division of 32-bit numbers by rw and by rw*rh as multiply-high + shift (host-computed magic
    // numbers, fastdiv below): the wavefront kernels turn slot ids into (sample, pixel) for every path
    uint64_t m_rw, m_nl;
};

// floor(x / d) for any 32-bit x: with m = floor(2^64 / d) + 1 the product's high half is exact for every
// x < 2^32 (the error term x / 2^64 * d stays below 1 / d).  d = 1 needs no magic.
__host__ __device__ inline uint32_t fastdiv(uint32_t x, uint32_t d, uint64_t m) {
    if (d == 1) return x;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(m, (uint64_t)x);
#else
    return (uint32_t)(((unsigned __int128)m * x) >> 64);
#endif
}
inline uint64_t fastdiv_magic(uint32_t d) { return d <= 1 ? 0 : (uint64_t)(~0ull / d) + 1; }

struct DeviceCounters {      // 64-bit accumulators in device memory
    unsigned long long rays, samples, box_tests, tri_tests, mesh_hits, env_lookups;
    unsigned long long trav_box_tests, trav_tri_tests;   // the share of box_tests / tri_tests counted inside k_wf_ext launches
#if defined(HRT_EXT_PROFILE) || defined(HRT_SHADE_PROFILE) || defined(HRT_STEP_PROFILE)       // experiments (tests/tools/ext_profile_run.py): where the lanes of k_wf_ext are, phase by phase
    unsigned long long prof[12];
#endif
};

// Number of set bits of a wave mask BELOW the calling lane (v_mbcnt_lo/hi: no per-lane 64-bit mask to keep in registers).
__device__ inline unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ inline unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool STATS>
__global__ __launch_bounds__(HRT_BLOCK) void k_pathtrace(DScene sc, hrt_camera cam, hrt_params pr, RenderMap map,
                                                         float* __restrict__ out, DeviceCounters* counters,
                                                         unsigned* work_counter) {
    __shared__ int s_stack[HRT_STACK_DEPTH * HRT_BLOCK];
    __shared__ __attribute__((aligned(16))) uint32_t s_tables[HRT_TABLE_LDS_BYTES / 4];
    int* stack = s_stack + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    stage_tables(sc, s_tables);

    // lane state
    int out_index = -1;          // local pixel (row-major in the region); -1 = lane holds no pixel
    int px = 0, py = 0;          // absolute pixel
    uint32_t pidx = 0;
    int s = 0;
    bool new_sample = false;
    bool exhausted = false;
    vec3 sum(0.0f);
    PathState ps; ps.o = vec3(0.0f); ps.d = vec3(0.0f); ps.atten = vec3(1.0f); ps.result = vec3(0.0f); ps.bounce = 0;
    PathCounters pc; pc.rays = 0; pc.samples = 0; pc.mesh_hits = 0; pc.env_lookups = 0; pc.bvh.box_tests = 0; pc.bvh.tri_tests = 0;

    for (;;) {
        // ---- refill: lanes without a pixel draw work items (one atomic per wave)
        const bool need = (out_index < 0) && !exhausted;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask) {
            const int leader = __ffsll((long long)need_mask) - 1;
            unsigned base = 0;
            if ((int)lane == leader) base = atomicAdd(work_counter, (unsigned)__popcll(need_mask));
            base = __shfl(base, leader, 64);
            if (need) {
                const unsigned item = base + (unsigned)__popcll(need_mask & ((1ull << lane) - 1ull));
                if (item >= (unsigned)map.total_items) {
                    exhausted = true;
                } else {
                    const int tile = (int)(item >> 6), within = (int)(item & 63u);
                    const int tx = tile % map.tiles_x, ty = tile / map.tiles_x;
                    const int lx = tx * 8 + (within & 7), ly = ty * 8 + (within >> 3);
                    if (lx < map.rw && ly < map.rh) {
                        if (map.mode == 0) { px = map.x0 + lx; py = map.y0 + ly; }
                        else { const int b = ly / map.R; px = lx; py = (b * map.G + map.rank) * map.R + (ly - b * map.R); }
                        out_index = ly * map.rw + lx;
                        pidx = (uint32_t)(py * pr.width + px);
                        s = 0; new_sample = true; sum = vec3(0.0f);
                    }
                }
            }
        }
        if (__ballot((out_index >= 0) || !exhausted) == 0ull) break;

        if (out_index >= 0) {
            rng_ctx ctx; ctx.seed_lo = pr.seed_lo; ctx.seed_hi = pr.seed_hi; ctx.pixel = pidx; ctx.sample = (uint32_t)s; ctx.bounce = 0;
            if (new_sample) { path_begin(cam, pr, px, py, ctx, ps); new_sample = false; pc.samples++; }
            const bool ended = path_segment<STATS>(sc, pr, ctx, ps, stack, pc);
            if (ended) {
                sum += ps.result;     // main.cpp:123
                s++;
                new_sample = true;
                if (s >= pr.samples) {
                    const vec3 mean = sum / static_cast<float>(pr.samples);  // main.cpp:126
                    float* op = out + 3ull * (unsigned)out_index;
                    op[0] = mean.x; op[1] = mean.y; op[2] = mean.z;
                    out_index = -1;
                }
            }
        }
    }

    // ---- counters: one 64-bit atomic per wave per counter
    const unsigned r = wave_sum(pc.rays), sm = wave_sum(pc.samples);
    unsigned bt = 0, tt = 0, mh = 0, ev = 0;
    if (STATS) { bt = wave_sum(pc.bvh.box_tests); tt = wave_sum(pc.bvh.tri_tests); mh = wave_sum(pc.mesh_hits); ev = wave_sum(pc.env_lookups); }
    if (lane == 0) {
        atomicAdd(&counters->rays, (unsigned long long)r);
        atomicAdd(&counters->samples, (unsigned long long)sm);
        if (STATS) {
            atomicAdd(&counters->box_tests, (unsigned long long)bt);
            atomicAdd(&counters->tri_tests, (unsigned long long)tt);
            atomicAdd(&counters->mesh_hits, (unsigned long long)mh);
            atomicAdd(&counters->env_lookups, (unsigned long long)ev);
        }
    }
}

__global__ __launch_bounds__(256) void k_resolve(const float* __restrict__ rgb, long long n_pixels, uint8_t* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n_pixels; i += stride) {
        uint8_t q[3];
        film_resolve(vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]), q);
        out[3 * i] = q[0]; out[3 * i + 1] = q[1]; out[3 * i + 2] = q[2];
    }
}

__global__ __launch_bounds__(HRT_BLOCK) void k_closest_hit(DScene sc, hrt_params pr, long long n, const float* __restrict__ ro,
                                                           const float* __restrict__ rd, float t_min, float t_max,
                                                           uint32_t pixel0, hrt_hit* __restrict__ out) {
    __shared__ int s_stack[HRT_STACK_DEPTH * HRT_BLOCK];
    __shared__ __attribute__((aligned(16))) uint32_t s_tables[HRT_TABLE_LDS_BYTES / 4];
    int* stack = s_stack + threadIdx.x;
    stage_tables(sc, s_tables);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vec3 o(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]);
    const vec3 d(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
    rng_ctx ctx; ctx.seed_lo = pr.seed_lo; ctx.seed_hi = pr.seed_hi; ctx.pixel = pixel0 + (uint32_t)i; ctx.sample = 0; ctx.bounce = 0;
    DCounters cnt; cnt.box_tests = 0; cnt.tri_tests = 0;
    sc.stale_ff = 1;      // the hit RECORD always carries the inherited frontFace (renders track it only where a material reads it)
    const WorldHit wh = world_hit<false>(sc, o, d, t_min, t_max, pr.quirks, ctx, stack, cnt);
    hrt_hit h;
    memset(&h, 0, sizeof(h));
    h.prim = wh.prim; h.tri = -1;
    if (wh.prim >= 0) {
        DRec rec;
        hit_record(sc, wh, o, d, pr.quirks, t_min, rec);
        h.t = rec.t;
        h.tri = sc.prims[wh.prim].kind == HRT_PRIM_MESH ? sub_tri(wh.sub) : -1;
        h.front_face = rec.frontFace ? 1 : 0;
        h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
        h.normal[0] = rec.normal.x; h.normal[1] = rec.normal.y; h.normal[2] = rec.normal.z;
        h.u = rec.u; h.v = rec.v;
    }
    out[i] = h;
}

__global__ void k_math_probe(int op, long long n, const float* __restrict__ in, const float* __restrict__ in2, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
        case 0: out[i] = gsin(in[i]); break;
        case 1: out[i] = gcos(in[i]); break;
        case 2: out[i] = gacos(in[i]); break;
        case 3: out[i] = gatan2(in2[i], in[i]); break;
        case 4: out[i] = glog(in[i]); break;
        case 5: {
            u32x4 r = philox4x32_10(f2u(in[4 * i]), f2u(in[4 * i + 1]), f2u(in[4 * i + 2]), f2u(in[4 * i + 3]), f2u(in2[2 * i]),
                                    f2u(in2[2 * i + 1]));
            out[4 * i] = u2f(r.x); out[4 * i + 1] = u2f(r.y); out[4 * i + 2] = u2f(r.z); out[4 * i + 3] = u2f(r.w);
            break;
        }
        // plain IEEE operations on awkward operands (denormal products, quotients, roots): what the compiler options and the
        // hardware mode registers make of them must be what the CPU makes of them (tests/test_gpu_parity.py)
        case 6: out[i] = in[i] * in2[i]; break;
        case 7: out[i] = in[i] / in2[i]; break;
        case 8: out[i] = sqrtf(in[i]); break;
        case 9: out[i] = in[i] + in2[i]; break;
        case 10: out[i] = fmaf(in[i], in2[i], in2[i]); break;
        case 11: out[i] = gsin_wide(in[i]); break;
        default: out[i] = 0.0f;
    }
}

// ===================================================================== wavefront pipeline (default path)
// The same render() as k_pathtrace, organised for wave density instead of per-lane persistence:
//   * every (pixel, sample) of the batch owns a SLOT (the 288 GB of HBM make a whole 640x640x100 frame
//     = 41 M concurrent paths affordable), so samples run in parallel; k_wf_reduce adds each pixel's
//     samples in sample order (main.cpp:118-126), so the fp32 sum keeps the reference's order;
//   * one ROUND = one path segment of every live path:
//       k_wf_ext    persistent BVH traversal of the rays that passed the root-box filter
//       k_wf_shade  remaining analytic prims, hitRecord, emitted/scatter (main.cpp:46-76); survivors get
//                   their NEXT segment prepared in the same kernel: analytic prims in front of the first
//                   mesh (list order), mesh-space ray, root-box filter, traversal record
//     (k_wf_gen does that preparation for the camera rays; scenes with several meshes run k_wf_pre +
//      k_wf_ext once more per further mesh, in world-list order);
//   * path state is PHYSICALLY COMPACTED every round (ping-pong buffers): a survivor writes its 64-byte
//     state to the next free position of its task's segment, so every kernel reads and writes dense,
//     coalesced records whatever fraction of the paths is still alive;
//   * lists are TASK-SEGMENTED: slots are cut into tasks of T consecutive positions, a task is always
//     handled by one wave, and compaction inside a task uses ballots only.  No append touches a global
//     atomic (one address sustains only ~88 atomics/us on this chip: MI355X_MICROARCH.md "dequeue"); the
//     traversal kernel pulls whole tasks, one atomic each.
// Everything is driven by DEVICE-side counts: the host enqueues max_depth rounds with no synchronisation.
struct WfBuf {
    // live-path state, two copies (read parity r & 1, written parity (r + 1) & 1), indexed by POSITION:
    float4* S0[2];   // o.xyz, d.x
    float4* S1[2];   // d.yz, attenuation.xy
    float4* S2[2];   // the hit so far: closest t, prim (bits), sub = triangle / box side (bits) | slot (bits).  The traversal
                     // kernel's result is ONE 12-byte store into this record (three scattered 4-byte stores into three records
                     // cost 204 MB of HBM writes per launch for ~14 MB of results: profiles/r01_v4)
    float* S3[2];    // attenuation.z
    float4* S4;      // scenes that track the stale frontFace only (wf_store_hit): t, prim, sub of the hit a wrapper-less mesh hit inherits it from
    // The running `result` of main.cpp:41 is NOT carried: Material::emitted is non-zero only for DiffuseLight
    // (material.h:67-70, 101-104), which never scatters (material.h:96-99), so result is still exactly 0 when a
    // path reaches its last segment and `0 + atten * x` is exact.
    float4 *E0, *E1, *E2, *E3;   // traversal records at [k*T, k*T + qn[k]): o' tmax | d' position | sX sY sZ kZ | 1/d'
    float4* rad;                 // radiance of the finished path of each SLOT
    unsigned* live;              // per task: live paths, at positions [k*T, k*T + live[k])
    unsigned* qn;                // per task: rays queued for the current mesh
    // Rays of quirk Q-4 that must walk the REFERENCE's tree (hrt_device.h q4_risky / ref_walk) are kept out of that queue:
    // their records (E0..E2) go to the TOP of the task's segment, at (k+1)*T - 1 downwards, rn[k] of them (qn + rn <= live
    // <= T, so the two never meet; the arrays are allocated to a whole number of tasks).
    unsigned* rn;
    // ... and the tasks that have such rays are listed for the traversal launch that will walk them: HRT_REF_GROUPS lists
    // (task t goes to list t % HRT_REF_GROUPS, ref_cap entries each: {task, rn}); a launch's counter block holds the lists'
    // lengths (words [0, 256)) and "next entry" counters (words [256, 512)).  ref_prod: the block of the launch the rays
    // are prepared FOR (written by k_wf_gen / k_wf_pre / k_wf_shade), ref_cons: the block of this launch (k_wf_ext).
    uint2* ref_list;
    unsigned ref_cap;
    unsigned* ref_prod;
    unsigned* ref_cons;
    // Segment counts, one cell per k_wf_shade wave, folded into DeviceCounters::rays by k_wf_reduce: one atomic per
    // wave on the one counter costs 8192 same-address atomics = 93 us per launch (~88 per us, MI355X_MICROARCH
    // "dequeue"), which was the floor of every round of a small tile (one rank's share of an 8-GPU frame).
    unsigned long long* wave_rays;
    unsigned n_wave_rays;
    unsigned* task_ctr;          // THIS launch's HRT_TASK_GROUPS "next task" counters (zeroed once per batch)
    unsigned n_groups;           // min(HRT_TASK_GROUPS, waves of this launch): every group has a wave
    unsigned pull_k;             // tasks per pull
    unsigned group_q, group_r;   // n_tasks / n_groups, n_tasks % n_groups
    unsigned T, n_tasks;
};
// Task ownership.  Tasks differ in cost by orders of magnitude (a run of pixels under the mesh vs. a run of sky), so a
// static wave -> task map leaves most waves idle while a few finish: plain striding (task = wave + i * n_waves) even
// resonates with the film -- n_waves x T is a whole number of images for power-of-two films (8192 waves x 4096 slots =
// 8 x 2048^2), a wave then gets the same image region in every turn (C5 bust: traversal 307 -> 470 ms per frame when the
// batches grew until T hit 4096).  One global "next task" counter is no answer either: same-address atomics retire at
// ~88 per us on MI355X, 16000 tasks + 6000 waves = 0.25 ms per launch.  So: HRT_TASK_GROUPS counters; group g owns the
// tasks g, g + G, g + 2G, ... (251 is prime: every group samples the whole film whatever its size) and its waves
// (wave mod G == g, ~24 of them) pull from the group's counter -- balanced within a group by the pulls, across groups by
// the interleaving, and ~100 atomics per address per launch.
// A pull takes `pull_k` of the group's tasks at once (host: so that a wave pulls ~4 times per launch however many tasks
// there are -- the atomic's round trip is ~2 us, which is the whole cost of a late round's nearly empty task), and a wave
// that can see it took the group's last task does not pull again to find out.
#define HRT_TASK_GROUPS 251
#define HRT_REF_GROUPS 251
// With no more tasks than waves (small batches: one rank's share of a multi-GPU frame) every wave simply takes the task of
// its own number (pull_k == 0): nothing to balance, and the atomic's round trip would be added to every launch.
struct TaskPuller { unsigned wave, g, next, end; };   // wave-uniform; the rest lives in the kernel arguments (registers are dear)
// (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes, and would keep the walk in VGPRs)
__device__ inline TaskPuller wf_task_puller(unsigned wave, unsigned n_groups) {
    TaskPuller p;
    p.wave = (unsigned)__builtin_amdgcn_readfirstlane((int)wave);
    p.g = p.wave % n_groups; p.next = 0; p.end = 0;
    return p;
}
#define HRT_TASK_PULLER(wave, n_groups) wf_task_puller((wave), (n_groups))
template <class WF>
__device__ inline bool wf_next_task(const WF& w, TaskPuller& p, unsigned lane, unsigned& task) {
    if (w.pull_k == 0) {                                        // one task per wave at most
        if (p.end != 0 || p.wave >= w.n_tasks) return false;
        p.end = 1; task = p.wave;
        return true;
    }
    if (p.next >= p.end) {
        const unsigned mine = w.group_q + (p.g < w.group_r ? 1u : 0u);   // tasks of this group (no division here: registers)
        if (p.end >= mine && p.end != 0) return false;         // this wave already holds the end of its group's list
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(&w.task_ctr[p.g], w.pull_k);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        p.next = k;
        p.end = k + w.pull_k < mine ? k + w.pull_k : mine;      // clipped: next < end  <=>  a task of this group
        if (p.end == 0) p.end = 1;                              // (group without tasks: the test above ends the walk)
        if (k >= mine) { p.next = p.end; return false; }
    }
    task = p.next * w.n_groups + p.g;
    ++p.next;
    return true;
}
// a per-task count: the same in every lane, but loaded through a vector load -- tell the compiler (loop bounds in SGPRs)
#define HRT_UNIFORM(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
#define HRT_FOR_MY_TASKS(task, w, wave, lane)                       \
    TaskPuller puller_ = HRT_TASK_PULLER((wave), (w).n_groups);     \
    for (unsigned task = 0; wf_next_task((w), puller_, (lane), task);)

struct WfScene {                 // world-list split points (host-computed)
    int first_mesh;              // index of the first HRT_PRIM_MESH, or n_prims when there is none
    int rest;                    // first prim after the last mesh (n_prims when there is no mesh)
    int has_mesh;
};

__device__ inline void slot_pixel(const RenderMap& map, unsigned lp, int& px, int& py) {
    const int ly = (int)fastdiv(lp, (unsigned)map.rw, map.m_rw);
    const int lx = (int)(lp - (unsigned)ly * (unsigned)map.rw);
    if (map.mode == 0) { px = map.x0 + lx; py = map.y0 + ly; }
    else { const int b = ly / map.R; px = lx; py = (b * map.G + map.rank) * map.R + (ly - b * map.R); }
}
__device__ inline rng_ctx slot_ctx(const hrt_params& pr, const RenderMap& map, unsigned slot, unsigned n_local, int s0, int bounce) {
    int px, py;
    const unsigned sl = fastdiv(slot, n_local, map.m_nl);
    slot_pixel(map, slot - sl * n_local, px, py);
    rng_ctx ctx; ctx.seed_lo = pr.seed_lo; ctx.seed_hi = pr.seed_hi;
    ctx.pixel = (uint32_t)(py * pr.width + px); ctx.sample = (uint32_t)(s0 + (int)sl); ctx.bounce = (uint32_t)bounce;
    return ctx;
}

// The wave-level stage functions are shared by several kernels (k_wf_shade / k_wf_tail ...): they must be INLINED into each --
// once wf_shade_task grew past the inliner's budget it became a real call, its by-reference state went to scratch
// (600 bytes per lane) and k_wf_shade took 58 instead of 21 ms per frame (tests/test_kernel_resources.py watches for it).
#define HRT_WAVE_FN __attribute__((always_inline)) inline
// Preparation of one segment: analytic prims [p0, p1) in list order (closest-so-far semantics of
// hittableList.cpp:12-19), then the ray in the space of mesh prim `mesh_prim` and the root-box filter.
// Returns 0: nothing to traverse, 1: queue the ray for the culling traversal, 2: queue it for the walk of the reference's tree.
#define HRT_ENQ_NONE 0
#define HRT_ENQ_TRAVERSE 1
#define HRT_ENQ_REFWALK 2
template <bool STATS>
__device__ HRT_WAVE_FN int wf_prepare(const DScene& sc, const hrt_params& pr, int p0, int p1, int mesh_prim, vec3 o, vec3 d,
                                  const rng_ctx& ctx, float& closest, int& prim, int& sub, MeshRay& mr, unsigned& n_culled) {
    prims_range_hit(sc, p0, p1, o, d, pr.t_min, pr.quirks, ctx, closest, prim, sub);
    if (mesh_prim < 0) return HRT_ENQ_NONE;
    const auto& mp = uniform_table(sc.prims)[mesh_prim];
    const auto& mesh = uniform_table(sc.meshes)[mp.mesh];
    vec3 lo = o, ld = d;
    for (int k = 0; k < mp.n_xforms; ++k) xf_apply(mp.xf[k], lo, ld, pr.quirks);
    float4 grid_o, grid_s;
    mesh_grid(sc, mp.mesh, grid_o, grid_s);
    mr = mesh_ray_setup(lo, ld, pr.quirks, grid_o, grid_s);
    if (q4_risky(mr.tr, ld, pr.quirks, sc.q4_route_a2)) {
        // the first step of the reference's walk, taken here: its root box (bvh.cpp:71) turns most of these rays away
        const HRT_CONST_AS uint32_t* rm = uniform_table((const uint32_t*)sc.rmesh) + 4 * mp.mesh;
        if (rm[1] == 0) return HRT_ENQ_NONE;
        const HRT_CONST_AS float* rb = uniform_table((const float*)sc.rnodes) + 8ull * rm[0];
        float4 bmn, bmx;
        bmn.x = rb[0]; bmn.y = rb[1]; bmn.z = rb[2]; bmn.w = 0.0f; bmx.x = rb[4]; bmx.y = rb[5]; bmx.z = rb[6]; bmx.w = 0.0f;
        return accept_box(bmn, bmx, lo, ld, pr.t_min, mesh_t_max(closest)) ? HRT_ENQ_REFWALK : HRT_ENQ_NONE;
    }
    const bool enq = root_may_hit(sc, mesh, mr, trav_t_lo(pr.t_min, pr.quirks), mesh_t_max(closest));
    if (STATS && !enq && mesh.node_count) n_culled++;
    return enq ? HRT_ENQ_TRAVERSE : HRT_ENQ_NONE;
}
__device__ inline void wf_store_record(const WfBuf& w, unsigned q, const MeshRay& mr, float closest, unsigned pos) {
    w.E0[q] = make_float4(mr.o.x, mr.o.y, mr.o.z, mesh_t_max(closest));   // what the traversal starts from: NaN -> +inf
    w.E1[q] = make_float4(mr.d.x, mr.d.y, mr.d.z, __uint_as_float(pos));
    w.E2[q] = make_float4(mr.tr.sX, mr.tr.sY, mr.tr.sZ, __int_as_float(mr.tr.kZ));
    w.E3[q] = make_float4(mr.idx, mr.idy, mr.idz, mr.reach);
}
// record of a ray for ref_walk: no culling constants
__device__ inline void wf_store_ref_record(const WfBuf& w, unsigned q, const MeshRay& mr, float closest, unsigned pos) {
    w.E0[q] = make_float4(mr.o.x, mr.o.y, mr.o.z, mesh_t_max(closest));
    w.E1[q] = make_float4(mr.d.x, mr.d.y, mr.d.z, __uint_as_float(pos));
    w.E2[q] = make_float4(mr.tr.sX, mr.tr.sY, mr.tr.sZ, __int_as_float(mr.tr.kZ));
}
// Queues the lanes' rays of one 64-position chunk: kind 1 upwards from qpos, kind 2 downwards from rtop - rcount.
__device__ HRT_WAVE_FN void wf_enqueue(const WfBuf& w, int kind, const MeshRay& mr, float closest, unsigned pos, unsigned long long lt,
                                       unsigned& qpos, unsigned rtop, unsigned& rcount) {
    const unsigned long long m = __ballot(kind == HRT_ENQ_TRAVERSE);
    if (kind == HRT_ENQ_TRAVERSE) wf_store_record(w, qpos + lanes_below(m), mr, closest, pos);
    qpos += (unsigned)__popcll(m);
    const unsigned long long mr2 = __ballot(kind == HRT_ENQ_REFWALK);
    if (mr2) {
        if (kind == HRT_ENQ_REFWALK) wf_store_ref_record(w, rtop - rcount - lanes_below(mr2), mr, closest, pos);
        rcount += (unsigned)__popcll(mr2);
    }
}
__device__ inline void wf_store_state(const WfBuf& w, int par, unsigned pos, const PathState& ps, float closest, unsigned slot, int prim, int sub) {
    w.S0[par][pos] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    w.S1[par][pos] = make_float4(ps.d.y, ps.d.z, ps.atten.x, ps.atten.y);
    w.S2[par][pos] = make_float4(closest, __int_as_float(prim), __int_as_float(sub), __uint_as_float(slot));
    w.S3[par][pos] = ps.atten.z;
}
// the hit-so-far part of a state record (what the stages between two shadings update)
// stale (wave-uniform): this is the hit of a mesh that stands in the world list without a wrapper, in a scene that tracks the
// stale frontFace (hrt_device.h WorldHit).  What the record holds so far IS the previous success of the list walk
// (hittableList.cpp:12-19: the stages run in list order): it is put aside in S4 as the source of the flag this hit inherits --
// unless it is such a mesh hit itself, which inherited its flag from the source already there.
__device__ inline void wf_store_hit(const WfBuf& w, int par, unsigned pos, int prim, int sub, float t, bool stale = false) {
    if (stale) {
        const float4 old = w.S2[par][pos];
        const int os = __float_as_int(old.z);
        if (__float_as_int(old.y) < 0) w.S4[pos] = make_float4(0.0f, __int_as_float(-1), __int_as_float(-1), 0.0f);
        else if (!(os >= 0 && (os & HRT_SUB_WRAPPERLESS))) w.S4[pos] = old;
        sub |= HRT_SUB_WRAPPERLESS;
    }
    float3 h; h.x = t; h.y = __int_as_float(prim); h.z = __int_as_float(sub);
    *(float3*)&w.S2[par][pos] = h;
}

// A task's ref-walk rays are announced to the launch that will walk them (lane 0 of the wave that prepared them).
__device__ inline void wf_ref_publish(const WfBuf& w, unsigned task, unsigned rcount) {
    const unsigned g = task % HRT_REF_GROUPS;
    const unsigned k = atomicAdd(&w.ref_prod[g], 1u);
    w.ref_list[(size_t)g * w.ref_cap + k] = make_uint2(task, rcount);
}
// Camera rays (main.cpp:115-123) of every slot of the batch + the preparation of their first segment.
template <bool STATS>
__global__ __launch_bounds__(256) void k_wf_gen(DScene sc, hrt_camera cam, hrt_params pr, RenderMap map, WfScene ws, unsigned n_local, int s0,
                                                unsigned n_slots, WfBuf w, DeviceCounters* counters) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned n_culled = 0;
    HRT_FOR_MY_TASKS(task, w, wave, lane) {
        const unsigned base = task * w.T;
        const unsigned n = base < n_slots ? min(w.T, n_slots - base) : 0u;
        unsigned qpos = base, rcount = 0;
        for (unsigned j0 = 0; j0 < n; j0 += 64) {
            const unsigned j = j0 + lane;
            int enq = HRT_ENQ_NONE;
            MeshRay mr;
            float closest = __builtin_huge_valf();
            const unsigned slot = base + j;
            if (j < n) {
                int px, py;
                const unsigned sl = fastdiv(slot, n_local, map.m_nl);
                slot_pixel(map, slot - sl * n_local, px, py);
                rng_ctx ctx; ctx.seed_lo = pr.seed_lo; ctx.seed_hi = pr.seed_hi;
                ctx.pixel = (uint32_t)(py * pr.width + px); ctx.sample = (uint32_t)(s0 + (int)sl); ctx.bounce = 0;
                PathState ps;
                path_begin(cam, pr, px, py, ctx, ps);
                int prim = -1, sub = -1;
                enq = wf_prepare<STATS>(sc, pr, 0, ws.first_mesh, ws.has_mesh ? ws.first_mesh : -1, ps.o, ps.d, ctx, closest, prim, sub, mr, n_culled);
                wf_store_state(w, 0, slot, ps, closest, slot, prim, sub);
            }
            wf_enqueue(w, enq, mr, closest, slot, lt, qpos, base + w.T - 1, rcount);
        }
        if (lane == 0) { w.live[task] = n; w.qn[task] = qpos - base; w.rn[task] = rcount; if (rcount) wf_ref_publish(w, task, rcount); }
    }
    if (STATS) {
        const unsigned c = wave_sum(n_culled);
        if (lane == 0 && c) atomicAdd(&counters->box_tests, 2ull * c);   // the root's two boxes were tested
    }
}

// ---- the three per-round stages as WAVE-LEVEL functions over one task; the kernels below (one launch per stage per
// ---- round, or k_wf_tail: every remaining round of a task in one go) only differ in how a wave comes by its tasks.

// Scenes with several meshes: analytic prims [p0, mesh_prim) + preparation for mesh prim `mesh_prim`.  Returns the number
// of rays queued for the traversal.
template <bool STATS>
__device__ HRT_WAVE_FN unsigned wf_pre_task(const DScene& sc, const hrt_params& pr, const RenderMap& map, unsigned n_local, int s0, int round, int par,
                                       int p0, int mesh_prim, const WfBuf& w, unsigned task, unsigned n, unsigned lane, unsigned long long lt,
                                       unsigned& n_culled, unsigned& rn_out) {
    const unsigned base = task * w.T;
    unsigned qpos = base, rcount = 0;
    for (unsigned j0 = 0; j0 < n; j0 += 64) {
        const unsigned pos = base + j0 + lane;
        int enq = HRT_ENQ_NONE;
        MeshRay mr;
        float closest = 0.0f;
        if (j0 + lane < n) {
            const float4 a = w.S0[par][pos], b = w.S1[par][pos], h = w.S2[par][pos];
            const vec3 o(a.x, a.y, a.z), d(a.w, b.x, b.y);
            closest = h.x;
            int prim = __float_as_int(h.y), sub = __float_as_int(h.z);
            const int prim0 = prim;
            const rng_ctx ctx = slot_ctx(pr, map, __float_as_uint(h.w), n_local, s0, round);
            enq = wf_prepare<STATS>(sc, pr, p0, mesh_prim, mesh_prim, o, d, ctx, closest, prim, sub, mr, n_culled);
            if (prim != prim0) wf_store_hit(w, par, pos, prim, sub, closest);
        }
        wf_enqueue(w, enq, mr, closest, pos, lt, qpos, base + w.T - 1, rcount);
    }
    rn_out = rcount;
    return qpos - base;
}
template <bool STATS>
__global__ __launch_bounds__(256) void k_wf_pre(DScene sc, hrt_params pr, RenderMap map, unsigned n_local, int s0, int round, int par,
                                                int p0, int mesh_prim, WfBuf w, DeviceCounters* counters) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned n_culled = 0;
    HRT_FOR_MY_TASKS(task, w, wave, lane) {
        unsigned rn;
        const unsigned qn = wf_pre_task<STATS>(sc, pr, map, n_local, s0, round, par, p0, mesh_prim, w, task, HRT_UNIFORM(w.live[task]), lane, lt, n_culled, rn);
        if (lane == 0) { w.qn[task] = qn; w.rn[task] = rn; if (rn) wf_ref_publish(w, task, rn); }
    }
    if (STATS) {
        const unsigned c = wave_sum(n_culled);
        if (lane == 0 && c) atomicAdd(&counters->box_tests, 2ull * c);
    }
}

// Scenes that track the stale frontFace (hrt_device.h WorldHit), after the round's last traversal: a path whose hit so far is
// that of a wrapper-less mesh and has a source for its flag (S4, wf_store_hit) gets the source's hitRecord built for that one
// bit; "false" is noted in the hit's sub-index (HRT_SUB_STALE_BACK), which is all k_wf_shade needs.  A stage of its own: a
// second hitRecord body inside k_wf_shade costs every scene 9 VGPRs and a spill.
__global__ __launch_bounds__(256) void k_wf_stale(DScene sc, hrt_params pr, int par, WfBuf w) {
    __shared__ __attribute__((aligned(16))) uint32_t s_tables[HRT_TABLE_LDS_BYTES / 4];
    stage_tables(sc, s_tables);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    HRT_FOR_MY_TASKS(task, w, wave, lane) {
        const unsigned base = task * w.T, n = HRT_UNIFORM(w.live[task]);
        for (unsigned j0 = 0; j0 < n; j0 += 64) {
            const unsigned pos = base + j0 + lane;
            if (j0 + lane >= n) continue;
            const float4 h = w.S2[par][pos];
            const int prim = __float_as_int(h.y), sub = __float_as_int(h.z);
            if (prim < 0 || sub < 0 || !(sub & HRT_SUB_WRAPPERLESS) || sc.lprims[prim].kind != HRT_PRIM_MESH) continue;
            const float4 src = w.S4[pos];
            WorldHit wh; wh.t = src.x; wh.prim = __float_as_int(src.y); wh.sub = __float_as_int(src.z); wh.s_prim = -1; wh.s_sub = -1; wh.s_t = 0.0f;
            if (wh.prim < 0) continue;
            HRT_BOUNDS(5, wh.prim, sc.n_prims);
            const float4 a = w.S0[par][pos], b = w.S1[par][pos];
            DRec rec;
            world_rec(sc, wh, vec3(a.x, a.y, a.z), vec3(a.w, b.x, b.y), pr.quirks, pr.t_min, rec);
            if (!rec.frontFace) w.S2[par][pos].z = __int_as_float(sub | HRT_SUB_STALE_BACK);
        }
    }
}

// BVH traversal of queued rays of one mesh prim by one wave.  `next_range(first, end)` hands the wave its next run of
// ray records (a task's queue) or returns false; finished lanes pull the next ray of the run (ballot + prefix count).
// While-while with postponed leaves: inner nodes are walked (two steps per wave vote: the ballots, counts, compare and
// branch are a third of a step's instructions) until at least leaf_num/64 of the busy lanes stand at a leaf -- waiting for
// ALL of them would run the loop at the pace of the slowest lane -- then the leaves are tested together; lanes still at an
// inner node sit the leaf phase out.
struct ExtMesh {                 // wave-uniform per-mesh constants of the traversal
    const uint4* nodes; const float4* tpos; const float4* tbox;
    float4 grid_o, grid_s;
    uint32_t node_count;
    bool stale;                  // see wf_store_hit
};
__device__ inline bool wf_mesh_stale(const DScene& sc, int mesh_prim, uint32_t quirks) {
    return stale_ff_tracked(sc, quirks) && uniform_table(sc.prims)[mesh_prim].n_xforms == 0;
}
__device__ inline ExtMesh wf_ext_mesh(const DScene& sc, int mesh_prim, uint32_t quirks) {
    ExtMesh m;
    const auto& mp = uniform_table(sc.prims)[mesh_prim];
    const auto& mesh = uniform_table(sc.meshes)[mp.mesh];
    m.nodes = sc.qnodes + 2ull * mesh.node_first;
    m.tpos = sc.tri_pos + 3ull * mesh.tri_first;
    m.tbox = sc.tri_box + 2ull * mesh.tri_first;
    mesh_grid(sc, mp.mesh, m.grid_o, m.grid_s);
    m.node_count = mesh.node_count;
    m.stale = wf_mesh_stale(sc, mesh_prim, quirks);
    return m;
}
struct RefMesh { const uint4* nodes; const float4* tris; const float4* tbox; uint32_t node_count, tri_count; bool stale; };   // the same for the reference-tree walks
__device__ inline RefMesh wf_ref_mesh(const DScene& sc, int mesh_prim, uint32_t quirks) {
    RefMesh m;
    const auto& mp = uniform_table(sc.prims)[mesh_prim];
    const auto& mesh = uniform_table(sc.meshes)[mp.mesh];
    const HRT_CONST_AS uint32_t* rm = uniform_table((const uint32_t*)sc.rmesh) + 4 * mp.mesh;
    m.nodes = sc.rnodes + 2ull * rm[0]; m.tris = sc.rtris + 3ull * rm[2]; m.node_count = rm[1];
    m.tbox = sc.tri_box + 2ull * mesh.tri_first; m.tri_count = mesh.tri_count;
    m.stale = wf_mesh_stale(sc, mesh_prim, quirks);
    return m;
}
// One LANE walks the reference's tree for the ray of record q (ref_walk: node by node, verbatim) and stores a hit like wf_ext_run does.
template <bool STATS>
__device__ HRT_WAVE_FN void wf_ref_one(const RefMesh& rm, const hrt_params& pr, int mesh_prim, int par, const WfBuf& w, unsigned q, DCounters& cnt) {
    const float4 e0 = w.E0[q], e1 = w.E1[q], e2 = w.E2[q];
    TriRay tr;
    const vec3 o(e0.x, e0.y, e0.z), d(e1.x, e1.y, e1.z);
    tr.o = o; tr.sX = e2.x; tr.sY = e2.y; tr.sZ = e2.z; tr.kZ = __float_as_int(e2.w);
    float t;
    const int tri = ref_walk<STATS>(rm.nodes, rm.tris, rm.node_count, o, d, tr, pr.t_min, e0.w, pr.quirks, t, cnt);
    if (tri >= 0) wf_store_hit(w, par, __float_as_uint(e1.w), mesh_prim, tri, t, rm.stale);
}
// One WAVE does the same for up to HRT_BFS_RAYS rays at once, breadth first.  A lane's walk is a chain of ~250 dependent
// node fetches, ~1 us each once the shading kernels' streams have swept the L2: longer than a late round's whole traversal
// launch, and every one of the 50 rounds waits for it.  What the walk computes is a FOLD over the triangles in the tree's
// depth-first order: a triangle that passes the t_max-independent part of ITriangle::hit (edge functions, determinant,
// sign of tScaled: triangle.cpp:98-105) is met with the t_max of that moment, is reached if the boxes above it let it
// through (bvh.cpp:71), and then shrinks t_max (bvh.cpp:75).  Of those boxes only the LOWEST can fail on account of
// t_max: a box is the exact union of what is below it and IEEE - and / are monotone, so a higher box's slab interval
// contains the lowest one's, and t_max only shrinks on the way down (up to the last-bit play of triangle.cpp:106-109).
// So: (1) the wave walks the tree level by level with the ray's initial t_max -- every (ray, node) pair of a level is one
// lane's box test, a level costs one fetch latency whatever its width, ~13 levels for the teapot -- and collects the
// candidate triangles; (2) lane r folds ray r's candidates in depth-first order, with the lowest node's box test (aabb.h:
// 26-39 as written: accept_box) done once per node with the t_max at the node's first candidate, and triangle.cpp:106-109
// as written.  The pairs live in the wave's share of the LDS traversal stack (`sb`, `words` of them), which is idle here.
// Rays with an exactly zero direction component (0/0 slabs are not monotone), and all rays of a group whose level or
// candidate list outgrows the LDS, take the lane's walk instead.
#define HRT_BFS_RAYS 8
#define HRT_BFS_CANDS 64
__device__ inline int& wave_lds(int* sb, unsigned i) { return sb[(i >> 6) * HRT_BLOCK + (i & 63u)]; }   // word i of the wave's share of stack[depth][256]
template <bool STATS>
__device__ HRT_WAVE_FN void wf_ref_bfs(const RefMesh& rm, const hrt_params& pr, int mesh_prim, int par, const WfBuf& w, unsigned n, unsigned q_mine,
                                       int* sb, unsigned words, unsigned lane, DCounters& cnt) {
    const unsigned off_cand = HRT_BFS_RAYS * 12u, off_f0 = off_cand + HRT_BFS_CANDS;
    const unsigned fcap = (words - off_f0) / 2u;
    const bool mine = lane < n;
    float4 e0 = make_float4(0, 0, 0, 0), e1 = e0, e2 = e0;
    bool alone = false;
    if (mine) {
        e0 = w.E0[q_mine]; e1 = w.E1[q_mine]; e2 = w.E2[q_mine];
        // (a component below 1e-30 has no finite reciprocal for the level walk's box test: treated like zero)
        alone = !(fabsf(e1.x) >= 1e-30f) || !(fabsf(e1.y) >= 1e-30f) || !(fabsf(e1.z) >= 1e-30f) || rm.node_count >= (1u << 26);
        const unsigned b = lane * 12u;
        wave_lds(sb, b + 0) = __float_as_int(e0.x); wave_lds(sb, b + 1) = __float_as_int(e0.y); wave_lds(sb, b + 2) = __float_as_int(e0.z);
        wave_lds(sb, b + 3) = __float_as_int(1.0f / e1.x); wave_lds(sb, b + 4) = __float_as_int(1.0f / e1.y); wave_lds(sb, b + 5) = __float_as_int(1.0f / e1.z);
        wave_lds(sb, b + 6) = __float_as_int(e2.x); wave_lds(sb, b + 7) = __float_as_int(e2.y); wave_lds(sb, b + 8) = __float_as_int(e2.z);
        wave_lds(sb, b + 9) = __float_as_int(e2.w);
        // t_max for the level walk: what an accepted t can exceed the t_max it was compared with (a few ulp per candidate)
        wave_lds(sb, b + 10) = __float_as_int(e0.w + fabsf(e0.w) * 2e-6f);
    }
    unsigned cur = off_f0, nxt = off_f0 + fcap, n_cur, n_cand = 0;
    {
        const bool go = mine && !alone;
        const unsigned long long m = __ballot(go);
        if (go) wave_lds(sb, cur + lanes_below(m)) = (int)(lane << 26);            // (ray, root)
        n_cur = (unsigned)__popcll(m);
    }
    bool overflow = false;
    while (n_cur && !overflow) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned n_next = 0;
        for (unsigned b = 0; b < n_cur && !overflow; b += 64) {
            const bool valid = b + lane < n_cur;
            unsigned r = 0, node = 0;
            bool pass = false;
            uint4 B; B.x = B.y = B.z = B.w = 0;
            vec3 o;
            if (valid) {
                const unsigned item = (unsigned)wave_lds(sb, cur + b + lane);
                r = item >> 26; node = item & 0x3ffffffu;
                const uint4 A = rm.nodes[2 * node];
                B = rm.nodes[2 * node + 1];
                const unsigned rb = r * 12u;
                o = vec3(__int_as_float(wave_lds(sb, rb + 0)), __int_as_float(wave_lds(sb, rb + 1)), __int_as_float(wave_lds(sb, rb + 2)));
                // The level walk only has to reach every node the reference's test (aabb.h:26-39: six IEEE divisions) would let
                // through with the ray's first t_max; the exact tests come in the fold.  So: (b - o) * (1 / d) -- within 2 ulp of
                // the quotient -- and a slack of 3e-7 of the two ends on the comparison.
                const float ix = __int_as_float(wave_lds(sb, rb + 3)), iy = __int_as_float(wave_lds(sb, rb + 4)), iz = __int_as_float(wave_lds(sb, rb + 5));
                const float ax = (__uint_as_float(A.x) - o.x) * ix, bx = (__uint_as_float(B.x) - o.x) * ix;
                const float ay = (__uint_as_float(A.y) - o.y) * iy, by = (__uint_as_float(B.y) - o.y) * iy;
                const float az = (__uint_as_float(A.z) - o.z) * iz, bz = (__uint_as_float(B.z) - o.z) * iz;
                const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), pr.t_min));
                const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), __int_as_float(wave_lds(sb, rb + 10))));
                pass = t_out - t_in >= -3e-7f * (fabsf(t_out) + fabsf(t_in)) - 1e-37f;
            }
            const bool inner = pass && (B.w & 0x80000000u);
            const unsigned long long mi = __ballot(inner);
            const unsigned n_in = (unsigned)__popcll(mi);
            if (n_next + 2u * n_in > fcap) { overflow = true; break; }
            if (inner) {
                const unsigned k = nxt + n_next + 2u * lanes_below(mi);
                wave_lds(sb, k) = (int)((r << 26) | (node + 1u));                         // left = the next node (bvh.cpp:74)
                wave_lds(sb, k + 1u) = (int)((r << 26) | (B.w & 0x3ffffffu));             // right
            }
            n_next += 2u * n_in;
            const bool leaf = pass && !inner;
            if (__ballot(leaf)) {
                TriRay tr;
                if (leaf) {
                    const unsigned rb = r * 12u;
                    tr.o = o; tr.sX = __int_as_float(wave_lds(sb, rb + 6)); tr.sY = __int_as_float(wave_lds(sb, rb + 7)); tr.sZ = __int_as_float(wave_lds(sb, rb + 8));
                    tr.kZ = wave_lds(sb, rb + 9);
                }
                for (unsigned k = 0; k < 2u; ++k) {
                    const unsigned p = (B.w >> 1) + k;
                    bool cand = false;
                    if (leaf && k <= (B.w & 1u)) {
                        const float4 q0 = rm.tris[3 * p + 0], q1 = rm.tris[3 * p + 1], q2 = rm.tris[3 * p + 2];
                        TriEval ev;
                        if (tri_eval(tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev))
                            cand = !((ev.det < 0 && ev.tScaled >= 0) || (ev.det > 0 && ev.tScaled <= 0));
                    }
                    const unsigned long long mc = __ballot(cand);
                    if (n_cand + (unsigned)__popcll(mc) > HRT_BFS_CANDS) { overflow = true; break; }
                    if (cand) wave_lds(sb, off_cand + n_cand + lanes_below(mc)) = (int)((r << 28) | p);
                    n_cand += (unsigned)__popcll(mc);
                }
            }
        }
        const unsigned t = cur; cur = nxt; nxt = t;
        n_cur = n_next;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (!mine) return;
    if (alone || overflow) { wf_ref_one<STATS>(rm, pr, mesh_prim, par, w, q_mine, cnt); return; }
    // (2) the fold over this lane's candidates in depth-first order
    TriRay tr;
    const vec3 o(e0.x, e0.y, e0.z), d(e1.x, e1.y, e1.z);
    tr.o = o; tr.sX = e2.x; tr.sY = e2.y; tr.sZ = e2.z; tr.kZ = __float_as_int(e2.w);
    float t_max = e0.w;
    int best = -1, last = -1;
    unsigned cur_node = 0xffffffffu;
    bool node_ok = false;
    for (;;) {
        unsigned p = 0xffffffffu;
        for (unsigned j = 0; j < n_cand; ++j) {
            const unsigned c = (unsigned)wave_lds(sb, off_cand + j);
            if ((c >> 28) == lane && (int)(c & 0xfffffffu) > last && (c & 0xfffffffu) < p) p = c & 0xfffffffu;
        }
        if (p == 0xffffffffu) break;
        last = (int)p;
        const float4 q0 = rm.tris[3 * p + 0], q1 = rm.tris[3 * p + 1], q2 = rm.tris[3 * p + 2];
        TriEval ev;
        if (!tri_eval(tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev)) continue;   // (cannot happen: it passed above)
        const unsigned ti = __float_as_uint(q0.w);
        const float4 bmn = rm.tbox[2 * ti], bmx = rm.tbox[2 * ti + 1];
        const unsigned node = __float_as_uint(bmn.w) >> 1;
        if (node != cur_node) { cur_node = node; node_ok = accept_box(bmn, bmx, o, d, pr.t_min, t_max); }
        if (!node_ok) continue;
        const float lim = t_max * ev.det;                                    // triangle.cpp:106-109
        if (ev.det < 0 && ev.tScaled < lim) continue;
        if (ev.det > 0 && ev.tScaled > lim) continue;
        const float t = ev.tScaled * (1 / ev.det);
        if (!(pr.quirks & HRT_Q2_TRI_NO_TMIN) && t < pr.t_min) continue;
        t_max = t;
        best = (int)ti;
    }
    if (best >= 0) wf_store_hit(w, par, __float_as_uint(e1.w), mesh_prim, best, t_max, rm.stale);
}
// The ref-walk rays of ONE task (k_wf_tail: by the wave that owns the task), HRT_BFS_RAYS at a time.
template <bool STATS>
__device__ HRT_WAVE_FN void wf_ref_task(const RefMesh& rm, const hrt_params& pr, int mesh_prim, int par, const WfBuf& w, unsigned task, unsigned rn,
                                        int* sb, unsigned words, unsigned lane, DCounters& cnt) {
    const unsigned top = (task + 1u) * w.T - 1u;
    for (unsigned done = 0; done < rn; done += HRT_BFS_RAYS) {
        const unsigned n = rn - done < HRT_BFS_RAYS ? rn - done : HRT_BFS_RAYS;
        wf_ref_bfs<STATS>(rm, pr, mesh_prim, par, w, n, top - (done + (lane < n ? lane : 0u)), sb, words, lane, cnt);
    }
}
// The ref-walk rays of ALL tasks, inside the k_wf_ext launch (beside the culling traversal, not after it: a launch of their
// own would add its latency to each of the 50 rounds), by every wave BEFORE it turns to its traversal tasks: in the early
// rounds these rays are 5 % of the launch's work, and since tasks are pulled dynamically the waves that walked some simply
// traverse less; in the late rounds the handful there is starts at once.  Wave w serves the lists w, w + waves, ... (mod
// HRT_REF_GROUPS; ~24 waves per list) and pulls one {task, rn} entry at a time until HRT_BFS_RAYS rays are together.  A wave
// looks before it pulls (plain loads of the list's length and counter): same-address atomics retire at ~88 per us.
template <bool STATS>
__device__ HRT_WAVE_FN void wf_ref_consume(const RefMesh& rm, const hrt_params& pr, int mesh_prim, int par, const WfBuf& w, unsigned wave, unsigned n_waves,
                                           int* sb, unsigned words, unsigned lane, DCounters& cnt) {
    for (unsigned g = wave % HRT_REF_GROUPS; g < HRT_REF_GROUPS; g += (n_waves < HRT_REF_GROUPS ? n_waves : HRT_REF_GROUPS)) {
        const unsigned len = HRT_UNIFORM(w.ref_cons[g]);
        if (len == 0 || HRT_UNIFORM(__atomic_load_n(&w.ref_cons[256 + g], __ATOMIC_RELAXED)) >= len) continue;
        unsigned top = 0, next = 0, end = 0;       // the entry being handed out: rays [next, end) at records top - i
        bool dry = false;
        for (;;) {
            unsigned have = 0, q = 0;
            while (have < HRT_BFS_RAYS) {
                if (next >= end) {
                    if (dry) break;
                    unsigned k = 0;
                    if (lane == 0) k = atomicAdd(&w.ref_cons[256 + g], 1u);
                    k = HRT_UNIFORM(k);
                    if (k >= len) { dry = true; break; }
                    const uint2 e = w.ref_list[(size_t)g * w.ref_cap + k];
                    top = (HRT_UNIFORM(e.x) + 1u) * w.T - 1u; next = 0; end = HRT_UNIFORM(e.y);
                }
                const unsigned take = end - next < HRT_BFS_RAYS - have ? end - next : HRT_BFS_RAYS - have;
                if (lane >= have && lane < have + take) q = top - (next + (lane - have));
                have += take; next += take;
            }
            if (!have) break;
            wf_ref_bfs<STATS>(rm, pr, mesh_prim, par, w, have, q, sb, words, lane, cnt);
        }
    }
}
#if defined(HRT_EXT_PROFILE) || defined(HRT_SHADE_PROFILE) || defined(HRT_STEP_PROFILE)
__device__ DeviceCounters* g_prof_counters;
#endif
#ifdef HRT_SHADE_PROFILE     // experiments: wave-cycles of k_wf_shade by phase (s_memtime at the phase boundaries)
#define HRT_SP_DECL unsigned long long sp_t_ = __builtin_readcyclecounter(), sp_acc_[6] = {0, 0, 0, 0, 0, 0}
#define HRT_SP_MARK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); sp_acc_[k] += n_ - sp_t_; sp_t_ = n_; }
#define HRT_SP_FLUSH() if (lane == 0) for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_prof_counters->prof[k_], sp_acc_[k_])
#else
#define HRT_SP_DECL
#define HRT_SP_MARK(k)
#define HRT_SP_FLUSH()
#endif
template <bool STATS, class NextRange>
__device__ HRT_WAVE_FN void wf_ext_run(const ExtMesh& em, const hrt_params& pr, int mesh_prim, int par, const WfBuf& w, int* stack, unsigned lane,
                                  unsigned long long lt, int leaf_num, DCounters& cnt, NextRange next_range) {
    const float t_lo = trav_t_lo(pr.t_min, pr.quirks);
    bool has = false;
    unsigned cur_pos = 0, cur_end = 0;   // wave-uniform: unread rays of the current run
    bool wave_done = false;              // wave-uniform: no run left
    MeshRay r;
    TravState ts;
    ts.cur = HRT_TRAV_DONE;
    unsigned pos = 0;
#ifdef HRT_EXT_PROFILE
    unsigned prof_[12] = {0};
#endif
#ifdef HRT_STEP_PROFILE
    // PER LANE (the stamps inside trav_inner are taken under the lanes' own control flow): load wait | box tests | descend / stack |
    // last stamp | votes + loop control (for a lane that sits a step out: the whole step) | wave steps seen | steps taken
    unsigned long long stp_[7] = {0, 0, 0, 0, 0, 0, 0};
#define HRT_STP_CTRL() { const unsigned long long n_ = __builtin_readcyclecounter(); stp_[4] += n_ - stp_[3]; stp_[3] = n_; stp_[5] += 1; }
#define HRT_STP_ARG , stp_
#else
#define HRT_STP_CTRL()
#define HRT_STP_ARG
#endif
    for (;;) {
        const unsigned long long need = __ballot(!has);
        if (need && !wave_done) {
            while (cur_pos >= cur_end && !wave_done)
                if (!next_range(cur_pos, cur_end)) wave_done = true;
            if (!wave_done) {
                const unsigned q = cur_pos + lanes_below(need);
                if (!has && q < cur_end) {
                    const float4 e0 = w.E0[q], e1 = w.E1[q], e2 = w.E2[q], e3 = w.E3[q];
                    r.o = vec3(e0.x, e0.y, e0.z); r.d = vec3(e1.x, e1.y, e1.z);
                    r.tr.o = r.o; r.tr.sX = e2.x; r.tr.sY = e2.y; r.tr.sZ = e2.z; r.tr.kZ = __float_as_int(e2.w);
                    r.idx = e3.x; r.idy = e3.y; r.idz = e3.z;
                    mesh_ray_grid(r, em.grid_o, em.grid_s, e3.w);
                    pos = __float_as_uint(e1.w);
                    ts.closest = e0.w; ts.best = -1;        // (wf_store_record wrote mesh_t_max(closest))
                    ts.selfhit = false; ts.self_order = 0xffffffffu; ts.self_tri = -1; ts.self_t = 0.0f;
                    ts.sp = 0;
                    ts.cur = em.node_count == 0 ? HRT_TRAV_DONE : 0;
                    has = true;
                }
                cur_pos += (unsigned)__popcll(need);
            }
        }
        if (__ballot(has) == 0ull) {
            if (wave_done) break;
            continue;
        }
#ifdef HRT_EXT_PROFILE
        prof_[0] += 1;
#endif
#ifdef HRT_STEP_PROFILE
        stp_[3] = __builtin_readcyclecounter();
#endif
        for (;;) {
            const unsigned long long m_in = __ballot(has && trav_at_inner(ts));
            if (!m_in) break;
            const int n_leaf = __popcll(__ballot(has && trav_at_leaf(ts)));
            if (n_leaf * 64 >= leaf_num * (n_leaf + __popcll(m_in))) break;
#ifdef HRT_EXT_PROFILE
            prof_[1] += 1; prof_[2] += (unsigned)__popcll(m_in); prof_[8] += (unsigned)__popcll(__ballot(has));
#endif
            HRT_STP_CTRL();
            if (has && trav_at_inner(ts)) trav_inner<STATS>(em.nodes, r, ts, t_lo, stack, cnt HRT_STP_ARG);
#ifdef HRT_EXT_PROFILE
            { const unsigned long long m2 = __ballot(has && trav_at_inner(ts)); if (m2) { prof_[1] += 1; prof_[2] += (unsigned)__popcll(m2); prof_[8] += (unsigned)__popcll(__ballot(has)); } }
#endif
            HRT_STP_CTRL();
            if (has && trav_at_inner(ts)) trav_inner<STATS>(em.nodes, r, ts, t_lo, stack, cnt HRT_STP_ARG);
        }
#ifdef HRT_EXT_PROFILE
        {
            const unsigned long long ml = __ballot(has && trav_at_leaf(ts));
            if (ml) { prof_[3] += 1; prof_[4] += (unsigned)__popcll(ml); prof_[6] += (unsigned)__popcll(__ballot(has));
                      prof_[5] += (unsigned)__popcll(__ballot(has && trav_at_leaf(ts) && (((unsigned)~ts.cur) & 7u) >= 1u)); }
        }
#endif
        if (has) {
            if (trav_at_leaf(ts)) trav_leaf<STATS>(em.tpos, em.tbox, r, ts, pr.t_min, pr.quirks, stack, cnt);
            if (ts.cur == HRT_TRAV_DONE) {
                float t;
                int tri = trav_result(ts, em.tpos, em.tbox, r, pr.t_min, t);
                if (tri >= 0 && trav_tie_overflow(ts)) tri |= HRT_SUB_TIE_UNSETTLED;      // three-way near-tie: world_rec settles it
                if (tri >= 0) wf_store_hit(w, par, pos, mesh_prim, tri, t, em.stale);
                has = false;
            }
        }
#ifdef HRT_EXT_PROFILE
        prof_[7] += (unsigned)__popcll(__ballot(!has));   // lanes without a ray at the end of an outer iteration
#endif
    }
#ifdef HRT_EXT_PROFILE
    if (lane == 0) for (int k = 0; k < 12; ++k) if (prof_[k]) atomicAdd(&g_prof_counters->prof[k], (unsigned long long)prof_[k]);
#endif
#ifdef HRT_STEP_PROFILE
    if (lane == 0) {      // lane 0 speaks for its wave: its own parts over the steps it took, the step length over all steps
        for (int k = 0; k < 3; ++k) atomicAdd(&g_prof_counters->prof[k], stp_[k]);
        atomicAdd(&g_prof_counters->prof[6], stp_[6]);
        atomicAdd(&g_prof_counters->prof[4], stp_[4] + stp_[0] + stp_[1] + stp_[2]); atomicAdd(&g_prof_counters->prof[5], stp_[5]);
    }
#endif
}
// One launch per round and mesh: persistent waves pull tasks (HRT_TASK_GROUPS) and their rays.
// DEPTH = entries of the per-lane LDS stack (>= the mesh's BVH depth, checked by the host): shallower trees
// leave room for more resident blocks per CU (20 or 24 entries: 6 blocks, the VGPR limit; 32 entries: 4).
template <bool STATS, int DEPTH>
__global__ __launch_bounds__(HRT_BLOCK) __attribute__((amdgpu_waves_per_eu(DEPTH <= 24 ? 6 : 5, DEPTH <= 24 ? 6 : 5))) void k_wf_ext(DScene sc, hrt_params pr, int mesh_prim, int par, WfBuf w,
                                                      DeviceCounters* counters, int leaf_num) {
    __shared__ int s_stack[DEPTH * HRT_BLOCK];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    DCounters cnt; cnt.box_tests = 0; cnt.tri_tests = 0;
    const unsigned wave = HRT_UNIFORM((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
#if defined(HRT_EXT_PROFILE) || defined(HRT_STEP_PROFILE)
    g_prof_counters = counters;
#endif
    if (pr.quirks & HRT_Q4_SHEAR_FROM_ORIGIN)
        wf_ref_consume<STATS>(wf_ref_mesh(sc, mesh_prim, pr.quirks), pr, mesh_prim, par, w, wave, (gridDim.x * blockDim.x) >> 6, s_stack + (threadIdx.x & ~63u), DEPTH * 64u, lane, cnt);
    {
        const ExtMesh em = wf_ext_mesh(sc, mesh_prim, pr.quirks);
        TaskPuller puller = HRT_TASK_PULLER(wave, w.n_groups);
        wf_ext_run<STATS>(em, pr, mesh_prim, par, w, s_stack + threadIdx.x, lane, lt, leaf_num, cnt, [&](unsigned& first, unsigned& end) {
            unsigned t;
            if (!wf_next_task(w, puller, lane, t)) return false;
            first = t * w.T; end = first + HRT_UNIFORM(w.qn[t]);
            return true;
        });
    }
    if (STATS) {
        const unsigned bt = wave_sum(cnt.box_tests), tt = wave_sum(cnt.tri_tests);
        if (lane == 0) {
            if (bt) { atomicAdd(&counters->box_tests, (unsigned long long)bt); atomicAdd(&counters->trav_box_tests, (unsigned long long)bt); }
            if (tt) { atomicAdd(&counters->tri_tests, (unsigned long long)tt); atomicAdd(&counters->trav_tri_tests, (unsigned long long)tt); }
        }
    }
}

// Paths that escaped (main.cpp:47-58) wait in a per-wave LDS queue until 64 of them can evaluate the background
// together: the environment lookup (normalize, atan2, acos, texel fetch: ~300 instructions) otherwise runs at
// the 10 % lane occupancy of "the lanes of this chunk that happened to miss".
#define HRT_MISSQ_CAP 128
struct MissQueue {            // one per wave; SoA rows of HRT_MISSQ_CAP entries: d.xyz, atten.xyz, slot
    float* f;                 // 6 rows
    unsigned* slot;           // 1 row
    unsigned count;           // wave-uniform
};
template <bool STATS>
__device__ inline void missq_flush(const DScene& sc, const WfBuf& w, MissQueue& q, unsigned lane, unsigned n, PathCounters& pc) {
    // the last n (<= 64) entries
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < n) {
        const unsigned e = q.count - n + lane;
        const vec3 d(q.f[0 * HRT_MISSQ_CAP + e], q.f[1 * HRT_MISSQ_CAP + e], q.f[2 * HRT_MISSQ_CAP + e]);
        const vec3 atten(q.f[3 * HRT_MISSQ_CAP + e], q.f[4 * HRT_MISSQ_CAP + e], q.f[5 * HRT_MISSQ_CAP + e]);
        const unsigned slot = q.slot[e];
        if (STATS && sc.ltexs[sc.background_tex].kind == HRT_TEX_ENV) pc.env_lookups++;
        vec3 result(0.0f);
        result += atten * background_value(sc, d);       // path_shade's miss branch, same operations
        w.rad[slot] = make_float4(result.x, result.y, result.z, 0.0f);
    }
    q.count -= n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The rest of the segment (analytic prims behind the last mesh, main.cpp:46-76) for every live path of one task, and,
// for the survivors, the preparation of their next segment.  Survivors are written compacted, in order, to the other
// state copy.  n = live paths of the task; returns the survivors (live_out) and the rays queued for the first mesh (qn_out).
template <bool STATS>
__device__ HRT_WAVE_FN void wf_shade_task(const DScene& sc, const hrt_params& pr, const RenderMap& map, const WfScene& ws, unsigned n_local, int s0, int round,
                                     const WfBuf& w, unsigned task, unsigned n, unsigned lane, unsigned long long lt, MissQueue& mq,
                                     PathCounters& pc, unsigned& n_seg, unsigned& n_culled, unsigned& live_out, unsigned& qn_out, unsigned& rn_out) {
    const int par = round & 1, nxt = par ^ 1;
    const unsigned base = task * w.T;
    unsigned out = base, qpos = base, rcount = 0;
    HRT_SP_DECL;
    for (unsigned j0 = 0; j0 < n; j0 += 64) {
        const unsigned pos = base + j0 + lane;
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
        float az = 0.0f;
        if (j0 + lane < n) { a = w.S0[par][pos]; b = w.S1[par][pos]; c = w.S2[par][pos]; az = w.S3[par][pos]; }
        bool alive = false;
        PathState ps;
        unsigned slot = 0;
        rng_ctx ctx; ctx.seed_lo = 0; ctx.seed_hi = 0; ctx.pixel = 0; ctx.sample = 0; ctx.bounce = 0;
        bool missed = false;
        WorldHit wh; wh.prim = -1; wh.sub = -1; wh.t = 0.0f; wh.s_prim = -1; wh.s_sub = -1; wh.s_t = 0.0f;
        if (j0 + lane < n) {
            n_seg++;
            ps.o = vec3(a.x, a.y, a.z); ps.d = vec3(a.w, b.x, b.y);
            ps.atten = vec3(b.z, b.w, az); ps.result = vec3(0.0f); ps.bounce = round;
            slot = __float_as_uint(c.w);
            float closest = c.x;
            int prim = __float_as_int(c.y), sub = __float_as_int(c.z);
            ctx = slot_ctx(pr, map, slot, n_local, s0, round);
            prims_range_hit(sc, ws.rest, sc.n_prims, ps.o, ps.d, pr.t_min, pr.quirks, ctx, closest, prim, sub);
            wh.prim = prim; wh.sub = sub; wh.t = closest;
            missed = prim < 0;
        }
        HRT_SP_MARK(0);
        {   // escaped paths: queue them; evaluate the background 64 at a time
            const unsigned long long mm = __ballot(missed);
            if (mm) {
                if (missed) {
                    const unsigned e = mq.count + lanes_below(mm);
                    mq.f[0 * HRT_MISSQ_CAP + e] = ps.d.x; mq.f[1 * HRT_MISSQ_CAP + e] = ps.d.y; mq.f[2 * HRT_MISSQ_CAP + e] = ps.d.z;
                    mq.f[3 * HRT_MISSQ_CAP + e] = ps.atten.x; mq.f[4 * HRT_MISSQ_CAP + e] = ps.atten.y; mq.f[5 * HRT_MISSQ_CAP + e] = ps.atten.z;
                    mq.slot[e] = slot;
                }
                mq.count += (unsigned)__popcll(mm);       // <= 63 + 64 < HRT_MISSQ_CAP
                if (mq.count >= 64) missq_flush<STATS>(sc, w, mq, lane, 64, pc);
            }
        }
        HRT_SP_MARK(1);
        if (j0 + lane < n && !missed) {
            const bool ended = path_shade<STATS>(sc, pr, ctx, ps, wh, pc);
            if (ended) w.rad[slot] = make_float4(ps.result.x, ps.result.y, ps.result.z, 0.0f);
            else {
                alive = true;
                // The state carries no `result`: while a path lives it is 0 -- main.cpp:66 adds attenuation * emitted, and
                // what scatters on emits nothing -- unless the attenuation is no longer finite (garbage scenes: inf * 0),
                // which makes that component of the pixel NaN for good.  A NaN attenuation says the same at the path's end.
                if (ps.result.x != ps.result.x) ps.atten.x = ps.result.x;
                if (ps.result.y != ps.result.y) ps.atten.y = ps.result.y;
                if (ps.result.z != ps.result.z) ps.atten.z = ps.result.z;
            }
        }
        const unsigned long long ma = __ballot(alive);
        HRT_SP_MARK(2);
        int enq = HRT_ENQ_NONE;
        MeshRay mr;
        float closest = __builtin_huge_valf();
        unsigned npos = 0;
        if (alive) {
            npos = out + lanes_below(ma);
            int prim = -1, sub = -1;
            ctx.bounce = (uint32_t)(round + 1);   // the next segment's draws (ConstantMedium::hit inside wf_prepare)
            enq = wf_prepare<STATS>(sc, pr, 0, ws.first_mesh, ws.has_mesh ? ws.first_mesh : -1, ps.o, ps.d, ctx, closest, prim, sub, mr, n_culled);
            wf_store_state(w, nxt, npos, ps, closest, slot, prim, sub);
        }
        HRT_SP_MARK(3);
        out += (unsigned)__popcll(ma);
        wf_enqueue(w, enq, mr, closest, npos, lt, qpos, base + w.T - 1, rcount);
        HRT_SP_MARK(4);
    }
    HRT_SP_FLUSH();
    live_out = out - base; qn_out = qpos - base; rn_out = rcount;
}
// what a shading wave adds to the device counters when it is done
template <bool STATS>
__device__ HRT_WAVE_FN void wf_shade_counters(const WfBuf& w, DeviceCounters* counters, unsigned wave, unsigned lane, unsigned n_seg, unsigned n_culled,
                                         const PathCounters& pc) {
    const unsigned seg = wave_sum(n_seg);
    if (lane == 0 && seg) w.wave_rays[wave] += (unsigned long long)seg;      // this wave's own cell: no contention
    if (STATS) {
        const unsigned mh = wave_sum(pc.mesh_hits), ev = wave_sum(pc.env_lookups), c = wave_sum(n_culled);
        if (lane == 0) {
            if (mh) atomicAdd(&counters->mesh_hits, (unsigned long long)mh);
            if (ev) atomicAdd(&counters->env_lookups, (unsigned long long)ev);
            if (c) atomicAdd(&counters->box_tests, 2ull * c);
        }
    }
}
#ifndef HRT_SHADE_WAVES
#define HRT_SHADE_WAVES 4   // waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif
template <bool STATS>
__global__ __launch_bounds__(256, HRT_SHADE_WAVES) void k_wf_shade(DScene sc, hrt_params pr, RenderMap map, WfScene ws, unsigned n_local, int s0, int round,
                                                  WfBuf w, DeviceCounters* counters) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __shared__ __attribute__((aligned(16))) uint32_t s_tables[HRT_TABLE_LDS_BYTES / 4];
    __shared__ float s_missq[4][7 * HRT_MISSQ_CAP];
#ifdef HRT_SHADE_PROFILE
    g_prof_counters = counters;
#endif
    stage_tables(sc, s_tables);
    MissQueue mq;
    mq.f = s_missq[threadIdx.x >> 6]; mq.slot = (unsigned*)(mq.f + 6 * HRT_MISSQ_CAP); mq.count = 0;
    unsigned n_seg = 0, n_culled = 0;
    PathCounters pc; pc.rays = 0; pc.samples = 0; pc.mesh_hits = 0; pc.env_lookups = 0; pc.bvh.box_tests = 0; pc.bvh.tri_tests = 0;
    HRT_FOR_MY_TASKS(task, w, wave, lane) {
        unsigned live, qn, rn;
        wf_shade_task<STATS>(sc, pr, map, ws, n_local, s0, round, w, task, HRT_UNIFORM(w.live[task]), lane, lt, mq, pc, n_seg, n_culled, live, qn, rn);
        if (lane == 0) { w.live[task] = live; w.qn[task] = qn; w.rn[task] = rn; if (rn) wf_ref_publish(w, task, rn); }
    }
    if (mq.count) missq_flush<STATS>(sc, w, mq, lane, mq.count, pc);
    wf_shade_counters<STATS>(w, counters, wave, lane, n_seg, n_culled, pc);
}

// Every remaining round [round0, rounds_end) of a task in one go, by the wave that pulled the task.  Rounds are a
// dependency chain per path, but the per-round launches also make every path wait for the slowest ray and the slowest
// chunk of the whole batch, twice per round, plus two launch gaps: once few paths are alive (or the batch is small: one
// rank's share of a multi-GPU frame) a round costs ~150 us whatever its size.  A task never exchanges paths with another
// task, so its wave can run traversal -> shading -> traversal ... on its own, at the pace of its own rays, while the other
// waves of the CU are in other stages of theirs.  Same device functions, same bits.  The stages of one wave communicate
// through the task's state and record arrays in HBM: a workgroup-scope fence (wait for the stores; all lanes share the
// CU's L1) orders them.
#define HRT_TAIL_MAX_MESHES 4
struct TailMeshes { int n; int prim[HRT_TAIL_MAX_MESHES]; };
// Residency is what this kernel lives on (a task's rounds are one long dependency chain; the more tasks run at once, the fewer
// wait for a wave): four blocks per CU with the 20- and 24-entry stacks -- 128 VGPRs (12 bytes of scratch) and a table
// budget cut to what is left of the LDS beside stack and miss queue (4 KB / 1.5 KB: the headline scene's tables take 1.3 KB;
// larger tables stay in global memory) -- three with the 32-entry stack.  One rank's 1/8 share of the headline frame:
// 10.9 -> 10.0 ms against three blocks with 12 KB tables, 9.7 ms with the switch into this kernel moved from round 20 to 8.
template <bool STATS, int DEPTH>
__global__ __launch_bounds__(256, DEPTH <= 24 ? 4 : 3) void k_wf_tail(DScene sc, hrt_params pr, RenderMap map, WfScene ws, TailMeshes tm, unsigned n_local, int s0, int round0,
                                                    int rounds_end, WfBuf w, DeviceCounters* counters, int leaf_num) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __shared__ int s_stack[DEPTH * 256];
    constexpr uint32_t TABLE_BYTES = DEPTH <= 20 ? 4096 : (DEPTH <= 24 ? 1536 : 4096);
    __shared__ __attribute__((aligned(16))) uint32_t s_tables[TABLE_BYTES / 4];
    __shared__ float s_missq[4][7 * HRT_MISSQ_CAP];
    stage_tables(sc, s_tables, TABLE_BYTES);
    MissQueue mq;
    mq.f = s_missq[threadIdx.x >> 6]; mq.slot = (unsigned*)(mq.f + 6 * HRT_MISSQ_CAP); mq.count = 0;
    unsigned n_seg = 0, n_culled = 0;
    PathCounters pc; pc.rays = 0; pc.samples = 0; pc.mesh_hits = 0; pc.env_lookups = 0; pc.bvh.box_tests = 0; pc.bvh.tri_tests = 0;
    DCounters cnt; cnt.box_tests = 0; cnt.tri_tests = 0;
    HRT_FOR_MY_TASKS(task, w, wave, lane) {
        unsigned live = HRT_UNIFORM(w.live[task]), qn = HRT_UNIFORM(w.qn[task]), rn = HRT_UNIFORM(w.rn[task]);     // as the last per-round launches left them
        for (int r = round0; r < rounds_end && live; ++r) {
            const int par = r & 1;
            for (int m = 0; m < tm.n; ++m) {
                if (m > 0) {
                    qn = wf_pre_task<STATS>(sc, pr, map, n_local, s0, r, par, tm.prim[m - 1] + 1, tm.prim[m], w, task, live, lane, lt, n_culled, rn);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                }
                if (rn) {
                    wf_ref_task<STATS>(wf_ref_mesh(sc, tm.prim[m], pr.quirks), pr, tm.prim[m], par, w, task, rn, s_stack + (threadIdx.x & ~63u), DEPTH * 64u, lane, cnt);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                }
                if (qn) {
                    const ExtMesh em = wf_ext_mesh(sc, tm.prim[m], pr.quirks);
                    bool given = false;
                    const unsigned first0 = task * w.T, end0 = first0 + qn;
                    wf_ext_run<STATS>(em, pr, tm.prim[m], par, w, s_stack + threadIdx.x, lane, lt, leaf_num, cnt, [&](unsigned& first, unsigned& end) {
                        if (given) return false;
                        given = true; first = first0; end = end0;
                        return true;
                    });
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                }
            }
            wf_shade_task<STATS>(sc, pr, map, ws, n_local, s0, r, w, task, live, lane, lt, mq, pc, n_seg, n_culled, live, qn, rn);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
        if (lane == 0) { w.live[task] = live; w.qn[task] = qn; w.rn[task] = rn; }
    }
    if (mq.count) missq_flush<STATS>(sc, w, mq, lane, mq.count, pc);
    wf_shade_counters<STATS>(w, counters, wave, lane, n_seg, n_culled, pc);
    if (STATS) {
        const unsigned bt = wave_sum(cnt.box_tests), tt = wave_sum(cnt.tri_tests);
        if (lane == 0) {
            if (bt) atomicAdd(&counters->box_tests, (unsigned long long)bt);
            if (tt) atomicAdd(&counters->tri_tests, (unsigned long long)tt);
        }
    }
}

// Per pixel: add the batch's samples IN SAMPLE ORDER (main.cpp:118-124); divide once all samples are in (main.cpp:126).
__global__ __launch_bounds__(256) void k_wf_reduce(const float4* __restrict__ rad, unsigned n_local, int chunk, int first_chunk, int last_chunk,
                                                   int spp, float* __restrict__ out, DeviceCounters* counters,
                                                   unsigned long long* __restrict__ wave_rays, unsigned n_wave_rays) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned lp = blockIdx.x * blockDim.x + threadIdx.x; lp < n_local; lp += stride) {
        vec3 sum(0.0f);
        if (!first_chunk) sum = vec3(out[3ull * lp], out[3ull * lp + 1], out[3ull * lp + 2]);
        for (int s = 0; s < chunk; ++s) {
            const float4 r = rad[(size_t)s * n_local + lp];
            sum += vec3(r.x, r.y, r.z);
        }
        if (last_chunk) sum = sum / static_cast<float>(spp);
        out[3ull * lp] = sum.x; out[3ull * lp + 1] = sum.y; out[3ull * lp + 2] = sum.z;
    }
    if (blockIdx.x == 0) {   // fold the per-wave segment counts of this batch's 50 k_wf_shade launches
        __shared__ unsigned long long total;
        if (threadIdx.x == 0) total = 0;
        __syncthreads();
        unsigned long long mine = 0;
        for (unsigned i = threadIdx.x; i < n_wave_rays; i += blockDim.x) { mine += wave_rays[i]; wave_rays[i] = 0; }
        if (mine) atomicAdd(&total, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (total) atomicAdd(&counters->rays, total);
            atomicAdd(&counters->samples, (unsigned long long)n_local * (unsigned long long)chunk);
        }
    }
}

// The progress counter of HRT_FLAG_PROGRESS (main.cpp:95-109): paths of this render call that have ended = `base` (batches
// already reduced) + the batch's slots - the paths still alive, written to host-mapped memory where the reporter reads it.
__global__ __launch_bounds__(256) void k_wf_progress(const unsigned* __restrict__ live, unsigned n_tasks, unsigned long long base_plus_slots,
                                                     volatile unsigned long long* host_ctr) {
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (unsigned i = threadIdx.x; i < n_tasks; i += blockDim.x) mine += live[i];
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) { *host_ctr = base_plus_slots - total; __threadfence_system(); }
}
__global__ void k_set_progress(unsigned long long v, volatile unsigned long long* host_ctr) { *host_ctr = v; __threadfence_system(); }

// ---- multi-GPU film assembly (hrt_multi_render): the gathered stripes of all ranks -> film order, and back
// gathered: G shares of `share` floats; rank g's share holds its rows_g x W x 3 floats (rows in increasing absolute order).
__global__ __launch_bounds__(256) void k_unstripe(const float* __restrict__ gathered, float* __restrict__ film, int H, int W3, int R, int G, long long share) {
    const long long n = (long long)H * W3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / W3), x = (int)(i - (long long)row * W3);
        const int b = row / R, g = b % G, local = (b / G) * R + (row - b * R);     // block b belongs to rank b % G (hrt_stripe_row_index inverted)
        film[i] = gathered[(long long)g * share + (long long)local * W3 + x];
    }
}
__global__ __launch_bounds__(256) void k_restripe(const float* __restrict__ film, float* __restrict__ mine, int H, int W3, int R, int G, int rank, int rows) {
    const long long n = (long long)rows * W3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int local = (int)(i / W3), x = (int)(i - (long long)local * W3);
        const int b = local / R, row = (b * G + rank) * R + (local - b * R);
        mine[i] = film[(long long)row * W3 + x];
    }
}
__global__ __launch_bounds__(256) void k_preview_mean(const float* __restrict__ sums, float* __restrict__ mean, long long n, float samples_done) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) mean[i] = sums[i] / samples_done;
}

}  // namespace

// ===================================================================== host side of the ABI
namespace {
thread_local std::string g_err;

hrt_status fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return HRT_ERR_HIP;
}
hrt_status fail(hrt_status s, const std::string& msg) { g_err = msg; return s; }
}  // namespace
// (for the library's other translation unit, hrt_lbvh.hip; not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) void hrt_set_last_error(const char* msg) { g_err = msg; }
namespace {

#define HIPCHK(expr)                                         \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return fail_hip(_e, #expr);    \
    } while (0)

template <typename T>
hrt_status upload(T** dptr, const void* src, size_t bytes) {
    *dptr = nullptr;
    size_t alloc = bytes ? bytes : 16;
    hipError_t e = hipMalloc((void**)dptr, alloc);
    if (e != hipSuccess) { g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP; }
    if (bytes) HIPCHK(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
    return HRT_OK;
}

bool finite3(const float* p) { return p[0] == p[0] && p[1] == p[1] && p[2] == p[2]; }
}  // namespace

struct WfWorkspace {          // device workspace of the wavefront pipeline (grown on demand, kept with the scene)
    void* base = nullptr;
    size_t bytes = 0;
    size_t slots = 0;
    int depth = 0, n_mesh = 0;
    WfBuf buf{};
};

struct hrt_scene {
    int device = 0;
    int n_cus = 256;
    DScene ds{};
    std::vector<void*> allocs;
    std::vector<int> mesh_prims;   // indices of the HRT_PRIM_MESH entries of the world list, in list order
    std::vector<int> mesh_depths;  // BVH depth of each of them (selects the traversal kernel's stack size)
    int n_prims = 0;
    DeviceCounters* d_counters = nullptr;
    unsigned* d_work = nullptr;
    WfWorkspace wf;
    // events of launches not yet folded into kernel_ms / traversal_ms
    struct Pending { hipEvent_t a, b; };
    std::vector<Pending> pending, pending_trav;
    std::vector<hipEvent_t> event_pool;
    double kernel_ms = 0.0, traversal_ms = 0.0;
    uint64_t launches = 0, traversal_launches = 0;
    // HRT_FLAG_PROGRESS: paths ended so far, in host memory mapped into the device (h_ / d_ are the two views of one word),
    // and what the host adds to it: paths of finished batches of the running call, and the call's total
    volatile unsigned long long* h_progress = nullptr;
    unsigned long long* d_progress = nullptr;
    unsigned long long progress_base = 0;
    volatile unsigned long long progress_total = 0;
};

namespace {

// Structural validation: every index the kernels follow must be in range, the
// BVH must be a tree no deeper than the LDS stack (hrt_device.h), so that a
// malformed scene is refused here instead of faulting the GPU.
hrt_status validate(const hrt_flat_scene* f) {
    if (!f) return fail(HRT_ERR_INVALID, "flat scene is NULL");
    if (f->n_prims && !f->prims) return fail(HRT_ERR_INVALID, "prims is NULL");
    if (f->n_textures == 0 || !f->textures) return fail(HRT_ERR_INVALID, "scene needs at least the background texture");
    if (f->background_tex < 0 || (uint32_t)f->background_tex >= f->n_textures) return fail(HRT_ERR_INVALID, "background_tex out of range");
    for (uint32_t i = 0; i < f->n_textures; ++i) {
        const hrt_texture& t = f->textures[i];
        if (t.kind < HRT_TEX_SOLID || t.kind > HRT_TEX_ENV) return fail(HRT_ERR_INVALID, "texture kind out of range");
        if (t.kind == HRT_TEX_CHECKER) {
            if (t.even < 0 || (uint32_t)t.even >= f->n_textures || t.odd < 0 || (uint32_t)t.odd >= f->n_textures)
                return fail(HRT_ERR_INVALID, "checker child out of range");
        }
        // width == 0 or height == 0 is the reference's "image failed to load" texture (texture.cpp:56-57, 79-80: cyan); any
        // other size must be positive and lie inside its texel array (the sums below cannot wrap: w, h < 2^31, channels < 2^31)
        if (t.kind == HRT_TEX_IMAGE || t.kind == HRT_TEX_ENV) {
            if (t.width < 0 || t.height < 0) return fail(HRT_ERR_INVALID, "negative texture size");
            if (t.width > 0 && t.height > 0) {
                const bool env = t.kind == HRT_TEX_ENV;
                if (env && t.channels < 3) return fail(HRT_ERR_INVALID, "environment map needs >= 3 channels");
                const uint64_t have = env ? f->n_texels_f32 : f->n_texels_u8;
                const uint64_t per = (uint64_t)t.width * (uint64_t)t.height;          // < 2^62
                const uint64_t ch = env ? (uint64_t)t.channels : 3u;
                if (t.offset > have || per > (have - t.offset) / ch) return fail(HRT_ERR_INVALID, env ? "env texels out of range" : "image texels out of range");
                if (have && !(env ? (const void*)f->texels_f32 : (const void*)f->texels_u8)) return fail(HRT_ERR_INVALID, "texel array is NULL");
            }
        }
    }
    // tex_value (hrt_device.h) follows at most 4 CheckeredTextures before it reaches a leaf: deeper or cyclic nests are refused,
    // not rendered cyan (the reference recurses without bound: a cycle there is a stack overflow)
    for (uint32_t i = 0; i < f->n_textures; ++i) {
        if (f->textures[i].kind != HRT_TEX_CHECKER) continue;
        std::vector<std::pair<uint32_t, int>> st;
        st.push_back({i, 1});
        while (!st.empty()) {
            auto [t, depth] = st.back(); st.pop_back();
            if (f->textures[t].kind != HRT_TEX_CHECKER) continue;
            if (depth > 4) return fail(HRT_ERR_UNSUPPORTED, "checkered textures nested more than 4 deep (or cyclic)");
            st.push_back({(uint32_t)f->textures[t].even, depth + 1});
            st.push_back({(uint32_t)f->textures[t].odd, depth + 1});
        }
    }
    if (f->n_materials && !f->materials) return fail(HRT_ERR_INVALID, "materials is NULL");
    if (f->n_meshes && !f->meshes) return fail(HRT_ERR_INVALID, "meshes is NULL");
    if (f->n_nodes && !f->nodes) return fail(HRT_ERR_INVALID, "nodes is NULL");
    auto tex_ok = [&](int32_t t) { return t < 0 || (uint32_t)t < f->n_textures; };
    for (uint32_t i = 0; i < f->n_materials; ++i) {
        const hrt_material& m = f->materials[i];
        if (m.kind < HRT_MAT_LAMBERTIAN || m.kind > HRT_MAT_UVTEST) return fail(HRT_ERR_INVALID, "material kind out of range");
        if (!tex_ok(m.albedo.tex) || !tex_ok(m.s0.tex) || !tex_ok(m.s1.tex)) return fail(HRT_ERR_INVALID, "material texture out of range");
        if (m.kind == HRT_MAT_PBR && (m.mix_tex < 0 || (uint32_t)m.mix_tex >= f->n_textures)) return fail(HRT_ERR_INVALID, "pbr mix texture out of range");
    }
    for (uint32_t i = 0; i < f->n_meshes; ++i) {
        const hrt_mesh& m = f->meshes[i];
        if ((uint64_t)m.tri_first + m.tri_count > f->n_tris) return fail(HRT_ERR_INVALID, "mesh triangle range out of bounds");
        if ((uint64_t)m.node_first + m.node_count > f->n_nodes) return fail(HRT_ERR_INVALID, "mesh node range out of bounds");
        if (m.tri_count >= (1u << 28)) return fail(HRT_ERR_UNSUPPORTED, "mesh too large for the leaf encoding");
        if (m.tri_count > 0 && m.node_count == 0) return fail(HRT_ERR_INVALID, "mesh has triangles but no BVH nodes");
        // walk the tree: indices in range, every node reached once, depth bounded, every triangle covered
        if (m.node_count) {
            std::vector<uint8_t> seen(m.node_count, 0);
            std::vector<uint8_t> tri_seen(m.tri_count, 0);
            std::vector<std::pair<int32_t, int>> st;
            st.push_back({0, 1});
            while (!st.empty()) {
                auto [ni, depth] = st.back(); st.pop_back();
                if (ni < 0 || (uint32_t)ni >= m.node_count) return fail(HRT_ERR_INVALID, "BVH child index out of range");
                if (seen[ni]) return fail(HRT_ERR_INVALID, "BVH is not a tree (node reached twice)");
                seen[ni] = 1;
                if (depth > HRT_STACK_DEPTH) return fail(HRT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
                const hrt_bvh_node& n = f->nodes[m.node_first + ni];
                const int32_t ch[2] = {n.child0, n.child1};
                const bool empty[2] = {n.c0_min_x > n.c0_max_x, n.c1_min_x > n.c1_max_x};
                for (int c = 0; c < 2; ++c) {
                    if (empty[c]) continue;
                    if (ch[c] >= 0) st.push_back({ch[c], depth + 1});
                    else {
                        uint32_t enc = (uint32_t)~ch[c];
                        uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
                        if ((uint64_t)first + count > m.tri_count) return fail(HRT_ERR_INVALID, "BVH leaf triangle range out of bounds");
                        for (uint32_t k = 0; k < count; ++k) tri_seen[first + k] = 1;
                    }
                }
            }
            for (uint32_t k = 0; k < m.tri_count; ++k)
                if (!tri_seen[k]) return fail(HRT_ERR_INVALID, "BVH does not cover every triangle");
        }
    }
    for (uint32_t i = 0; i < f->n_prims; ++i) {
        const hrt_prim& p = f->prims[i];
        if (p.kind < HRT_PRIM_SPHERE || p.kind > HRT_PRIM_TRIANGLE) return fail(HRT_ERR_INVALID, "prim kind out of range");
        if (p.material < 0 || (uint32_t)p.material >= f->n_materials) return fail(HRT_ERR_INVALID, "prim material out of range");
        if (p.kind == HRT_PRIM_MESH && (p.mesh < 0 || (uint32_t)p.mesh >= f->n_meshes)) return fail(HRT_ERR_INVALID, "prim mesh out of range");
        if (p.kind == HRT_PRIM_MEDIUM) {
            if (p.boundary_kind != HRT_PRIM_SPHERE && p.boundary_kind != HRT_PRIM_BOX)
                return fail(HRT_ERR_UNSUPPORTED, "constant medium boundary must be a sphere or a box");
            // (any density is taken, as constantMedium.cpp:9 does: -1 / 0 = -inf simply never scatters, a negative one
            // scatters "before" the boundary; the arithmetic is the oracle's, tests/scene_helpers.py random_world)
        }
        if (p.n_xforms < 0 || p.n_xforms > HRT_MAX_XFORMS) return fail(HRT_ERR_INVALID, "too many instance wrappers");
        for (int k = 0; k < p.n_xforms; ++k)
            if (p.xf[k].kind < HRT_XF_TRANSLATE || p.xf[k].kind > HRT_XF_ROTATE_Y) return fail(HRT_ERR_INVALID, "wrapper kind out of range");
    }
    if (f->n_tris && (!f->tri_pos || !f->tri_nrm || !f->tri_uv)) return fail(HRT_ERR_INVALID, "triangle arrays missing");
    // vertex positions feed the culling-node packer and the traversal's index arithmetic: NaN / inf there is refused, not rendered
    for (uint64_t i = 0; i < (uint64_t)f->n_tris * 9; ++i)
        if (!std::isfinite(f->tri_pos[i])) return fail(HRT_ERR_INVALID, "non-finite vertex position (triangle " + std::to_string(i / 9) + ")");
    // ... and so is a mesh whose extent could overflow fp32: the culling grid needs 65535 x step finite (hrt_pack.h)
    for (uint64_t i = 0; i < (uint64_t)f->n_tris * 9; ++i)
        if (std::fabs(f->tri_pos[i]) > 8e37f) return fail(HRT_ERR_UNSUPPORTED, "vertex coordinate beyond +-8e37 (triangle " + std::to_string(i / 9) + ")");
    for (uint32_t i = 0; i < f->n_nodes; ++i) {
        const hrt_bvh_node& n = f->nodes[i];
        const float b[12] = {n.c0_min_x, n.c0_min_y, n.c0_min_z, n.c0_max_x, n.c0_max_y, n.c0_max_z, n.c1_min_x, n.c1_min_y, n.c1_min_z, n.c1_max_x, n.c1_max_y, n.c1_max_z};
        for (int c = 0; c < 2; ++c) {
            if (b[6 * c] > b[6 * c + 3]) continue;                      // empty child marker (+inf, -inf)
            for (int k = 0; k < 6; ++k)
                if (!std::isfinite(b[6 * c + k]) || std::fabs(b[6 * c + k]) > 8.1e37f) return fail(HRT_ERR_INVALID, "non-finite or out-of-range BVH box (node " + std::to_string(i) + ")");
        }
    }
    return HRT_OK;
}

// Depth of a (validated) mesh BVH: the deepest inner node, root = 1 (what validate() bounds by HRT_STACK_DEPTH).
int bvh_depth(const hrt_flat_scene* f, const hrt_mesh& m) {
    int deepest = 0;
    if (!m.node_count) return deepest;
    std::vector<std::pair<int32_t, int>> st;
    st.push_back({0, 1});
    while (!st.empty()) {
        auto [ni, depth] = st.back(); st.pop_back();
        deepest = std::max(deepest, depth);
        const hrt_bvh_node& n = f->nodes[m.node_first + ni];
        if (!(n.c0_min_x > n.c0_max_x) && n.child0 >= 0) st.push_back({n.child0, depth + 1});
        if (!(n.c1_min_x > n.c1_max_x) && n.child1 >= 0) st.push_back({n.child1, depth + 1});
    }
    return deepest;
}

hrt_status check_params(const hrt_params* p) {
    if (!p) return fail(HRT_ERR_INVALID, "params is NULL");
    if (p->width < 2 || p->height < 2) return fail(HRT_ERR_INVALID, "film must be at least 2x2 (main.cpp:120-121 divides by W-1, H-1)");
    if ((int64_t)p->width * p->height > (int64_t)1 << 30) return fail(HRT_ERR_UNSUPPORTED, "film larger than 2^30 pixels");
    // the megakernel deals pixels in 8x8 tiles: the PADDED pixel count must fit 31 bits too (a 2 x 2^29 film pads to 2^32)
    if (((int64_t)p->width + 7) / 8 * (((int64_t)p->height + 7) / 8) * 64 > (int64_t)0x7fffffff) return fail(HRT_ERR_UNSUPPORTED, "film too thin: its 8x8-tile padding exceeds 2^31 pixels");
    if (p->samples < 1) return fail(HRT_ERR_INVALID, "samples must be >= 1");
    if (p->max_depth < 1) return fail(HRT_ERR_INVALID, "max_depth must be >= 1");
    // the wavefront pipeline enqueues two launches and 3 KB of counters per round whether paths are left or not
    if (p->max_depth > 65536) return fail(HRT_ERR_UNSUPPORTED, "max_depth above 65536 (the reference's is 50, main.cpp:32)");
    if (!(p->t_min == p->t_min)) return fail(HRT_ERR_INVALID, "t_min is NaN");
    return HRT_OK;
}

hrt_status get_event(hrt_scene* sc, hipEvent_t* ev) {
    if (!sc->event_pool.empty()) { *ev = sc->event_pool.back(); sc->event_pool.pop_back(); return HRT_OK; }
    HIPCHK(hipEventCreate(ev));
    return HRT_OK;
}

hrt_status launch_megakernel(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, const RenderMap& map, float* d_out,
                             hipStream_t stream) {
    HIPCHK(hipMemsetAsync(sc->d_work, 0, sizeof(unsigned), stream));
    int blocks = (map.total_items + HRT_BLOCK - 1) / HRT_BLOCK;
    const int cap = sc->n_cus * 8;
    if (blocks > cap) blocks = cap;
    if (pr->flags & HRT_FLAG_STATS)
        hipLaunchKernelGGL(k_pathtrace<true>, dim3(blocks), dim3(HRT_BLOCK), 0, stream, sc->ds, *cam, *pr, map, d_out, sc->d_counters, sc->d_work);
    else
        hipLaunchKernelGGL(k_pathtrace<false>, dim3(blocks), dim3(HRT_BLOCK), 0, stream, sc->ds, *cam, *pr, map, d_out, sc->d_counters, sc->d_work);
    HIPCHK(hipGetLastError());
    return HRT_OK;
}

// Most (pixel, sample) slots one batch may hold.  Every batch pays ~50 rounds x 2 kernels of fixed latency, so films
// whose paths die early (open scenes: C4 shiny_teapot spends 22 batches at a 48 Mi cap, 141 ms; 2 batches, 91 ms) want
// the largest batch that fits: 184 B per slot, up to 60 % of the free HBM (288 GB per MI355X) and 2^31 slots.  The
// workspace is only ever as large as the batch needs (the headline frame: 41 M slots = 7.5 GB).
size_t wf_max_slots(const hrt_scene* sc) {
    if (const char* e = getenv("HRT_WF_MAX_SLOTS")) { long long v = atoll(e); if (v > 0) return (size_t)v; }
    size_t cap = (size_t)48 << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        free_b += sc->wf.bytes;                          // what we hold already can be re-used
        cap = std::max(cap, (size_t)((double)free_b * 0.6 / 184.0));
    }
    return std::min(cap, (size_t)1 << 31);
}

// "next task" counters: one block of 256 words per kernel launch of a batch (gen + per round: ext and pre per mesh, shade)
// (+ k_wf_stale in the scenes that have it)
size_t wf_counter_words(int depth, int n_mesh) { return (size_t)256 * (2 + (size_t)depth * (2 * (size_t)std::max(1, n_mesh) + 2)); }
// ... and of 512 words per traversal launch for its ref-walk lists (WfBuf::ref_prod / ref_cons), + one block nobody reads
size_t wf_ref_counter_words(int depth, int n_mesh) { return (size_t)512 * ((size_t)depth * (size_t)std::max(1, n_mesh) + 1); }

hrt_status wf_reserve(hrt_scene* sc, size_t slots, int depth) {
    WfWorkspace& w = sc->wf;
    const int n_mesh = (int)sc->mesh_prims.size();
    if (w.base && w.slots >= slots && w.depth >= depth && w.n_mesh == n_mesh) return HRT_OK;
    if (w.base) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(w.base); w = WfWorkspace(); }
    const size_t max_tasks = slots / 64 + 1;   // the smallest task HRT_WF_TASK_SIZE can ask for is 64 positions (the default is >= 256)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t f4 = al((slots + 4096) * sizeof(float4));   // to a whole number of tasks (T <= 4096): ref-walk records sit at the top of a task's segment
    const size_t i4 = al((slots + 4096) * sizeof(int));
    const size_t ctr_words = wf_counter_words(depth, n_mesh) + wf_ref_counter_words(depth, n_mesh);
    const size_t ref_cap = max_tasks / HRT_REF_GROUPS + 1;
    const size_t total = (sc->ds.stale_ff ? 12 : 11) * f4 + 2 * i4 + 3 * al(max_tasks * sizeof(unsigned)) + al(ctr_words * sizeof(unsigned)) + al(ref_cap * HRT_REF_GROUPS * sizeof(uint2)) +
                         al((size_t)sc->n_cus * 32 * sizeof(unsigned long long));                                      // 184 B per slot
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, total);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string("hipMalloc(wavefront workspace): ") + hipGetErrorString(e));
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* q = p; p += bytes; return q; };
    for (int k = 0; k < 2; ++k) {
        w.buf.S0[k] = (float4*)take(f4); w.buf.S1[k] = (float4*)take(f4); w.buf.S2[k] = (float4*)take(f4); w.buf.S3[k] = (float*)take(i4);
    }
    w.buf.E0 = (float4*)take(f4); w.buf.E1 = (float4*)take(f4); w.buf.E2 = (float4*)take(f4); w.buf.E3 = (float4*)take(f4);
    w.buf.rad = (float4*)take(f4);
    w.buf.S4 = sc->ds.stale_ff ? (float4*)take(f4) : nullptr;
    w.buf.live = (unsigned*)take(al(max_tasks * sizeof(unsigned)));
    w.buf.qn = (unsigned*)take(al(max_tasks * sizeof(unsigned)));
    w.buf.rn = (unsigned*)take(al(max_tasks * sizeof(unsigned)));
    w.buf.task_ctr = (unsigned*)take(al(ctr_words * sizeof(unsigned)));
    w.buf.ref_list = (uint2*)take(al(ref_cap * HRT_REF_GROUPS * sizeof(uint2)));
    w.buf.ref_cap = (unsigned)ref_cap;
    w.buf.n_wave_rays = (unsigned)sc->n_cus * 8u * 4u;           // k_wf_shade never runs more waves (task_blocks <= 8 per CU, 4 waves each)
    w.buf.wave_rays = (unsigned long long*)take(al((size_t)w.buf.n_wave_rays * sizeof(unsigned long long)));
    HIPCHK(hipMemset(w.buf.wave_rays, 0, (size_t)w.buf.n_wave_rays * sizeof(unsigned long long)));
    w.base = base; w.bytes = total; w.slots = slots; w.depth = depth; w.n_mesh = n_mesh;
    return HRT_OK;
}

// render() as the wavefront pipeline: see the comment above struct WfBuf.
// Samples [s_first, s_first + s_count) of pr->samples are added, in sample order, to the sums held in d_out
// (s_first == 0 starts them); the batch that reaches pr->samples divides (main.cpp:126).
hrt_status launch_wavefront(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, const RenderMap& map, float* d_out,
                            hipStream_t stream, int s_first, int s_count) {
    const unsigned n_local = (unsigned)map.rw * (unsigned)map.rh;
    const int D = pr->max_depth;
    const int n_mesh = (int)sc->mesh_prims.size();
    size_t cap = wf_max_slots(sc);
    const int s_end = s_first + s_count;
    int chunk = (int)std::min<size_t>((size_t)s_count, std::max<size_t>(1, cap / n_local));
    // (the largest batches that fit plus a remainder, NOT equal batches: C4 on one GPU takes 92.8 ms as 453 + 59 samples and
    //  97.1 ms as 256 + 256 -- a batch's rounds run the better the more paths they hold, round 3)
    hrt_status st;
    for (;;) {   // the memory estimate can be stale (other processes on the device): halve the batch on OOM
        const size_t slots = (size_t)n_local * chunk;
        if (slots >= ((size_t)1 << 32) - 4096) return fail(HRT_ERR_UNSUPPORTED, "tile too large for 32-bit slot ids");
        st = wf_reserve(sc, slots, D);
        if (st != HRT_ERR_OOM || chunk == 1) break;
        chunk = (chunk + 1) / 2;
    }
    if (st != HRT_OK) return st;
    WfBuf w = sc->wf.buf;
    WfScene ws;
    ws.has_mesh = n_mesh > 0;
    ws.first_mesh = n_mesh > 0 ? sc->mesh_prims.front() : sc->n_prims;
    ws.rest = n_mesh > 0 ? sc->mesh_prims.back() + 1 : sc->n_prims;
    const bool stats = (pr->flags & HRT_FLAG_STATS) != 0, timing = (pr->flags & HRT_FLAG_TIMING) != 0;
    const bool progress = (pr->flags & HRT_FLAG_PROGRESS) != 0 && sc->d_progress;
    int ext_per_cu_env = 0;                              // experiments: k_wf_ext blocks per CU
    if (const char* e = getenv("HRT_EXT_BLOCKS_PER_CU")) ext_per_cu_env = std::min(10, std::max(1, atoi(e)));
    // from which round on a task's remaining rounds run in one k_wf_tail launch (>= D: never)
    // Measured (tests/tools/stripe_scaling.py, sweep_env.py; round 2, k_wf_tail at four blocks per CU): on the full headline
    // frame and its half every switch point loses or ties (51.5 ms without, 51.5 at round 40, 52.4 at 24), on the 1/4 share
    // round 24 wins 4 % (18.5 -> 17.8 ms), on the 1/8 share rounds 2..8 win 11 % (10.9 -> 9.7 ms; 10.1 at round 20).
    // Tiny batches (previews: <= 512 Ki slots) are nothing but launch and wait: the tail from round 1 renders 64x64x4 in
    // 2.2 instead of 2.9 ms.
    const size_t batch_slots = (size_t)n_local * (size_t)chunk;
    int tail_round = batch_slots <= ((size_t)512 << 10) ? 1 : (batch_slots <= ((size_t)6 << 20) ? 8 : (batch_slots <= ((size_t)12 << 20) ? 24 : D));
    if (const char* e = getenv("HRT_WF_TAIL_ROUND")) tail_round = std::max(1, atoi(e));
    if (n_mesh > HRT_TAIL_MAX_MESHES) tail_round = D;
    const bool stale_ff = sc->ds.stale_ff && (pr->quirks & HRT_Q3_TRI_NO_FACE);
    if (stale_ff) tail_round = D;        // k_wf_tail has no stage for the stale frontFace (k_wf_stale): round by round
    tail_round = std::min(tail_round, D);
    int leaf_num = 48;                                   // k_wf_ext: start the leaf phase when >= 48/64 of the busy lanes wait at a leaf
    if (const char* e = getenv("HRT_EXT_LEAF_NUM")) leaf_num = atoi(e);
    leaf_num = std::min(64, std::max(1, leaf_num));      // >= 1: with no lane at a leaf the inner loop must go on

    for (int s0 = s_first; s0 < s_end; s0 += chunk) {
        const int c = std::min(chunk, s_end - s0);
        const unsigned n_slots = n_local * (unsigned)c;
        // Task size (positions per task, a multiple of 64).  Large batches: ~64 tasks per CU, so that the strided static
        // ownership averages over several tasks per wave.  Small batches (one rank's share of a multi-GPU frame): no
        // more tasks than the ~4096 waves k_wf_shade keeps resident, but up to 1280 positions each -- in the late rounds
        // a task's survivors then still fill whole 64-lane chunks and a round costs ~2 instead of ~4 dependent chunk
        // passes (measured on the 1/8 share of the headline frame: 11.4 -> 10.2 ms; tests/tools/stripe_scaling.py).
        auto round64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
        const size_t t_large = round64(n_slots / ((size_t)sc->n_cus * 64));
        const size_t t_small = std::min<size_t>(1280, std::max<size_t>(256, round64(n_slots / ((size_t)sc->n_cus * 16))));
        size_t T = std::min<size_t>(4096, std::max(t_large, t_small));
        if (const char* e = getenv("HRT_WF_TASK_SIZE")) T = std::min<size_t>(4096, std::max<size_t>(64, round64((size_t)atoi(e))));   // experiments
        w.T = (unsigned)T;
        w.n_tasks = (unsigned)((n_slots + T - 1) / T);
        const int task_blocks = (int)std::min<size_t>(((size_t)w.n_tasks + 3) / 4, (size_t)sc->n_cus * 8);   // 4 waves = 4 tasks per block
        // How many task groups.  Slot = sample x n_local + pixel, so the tasks of one image region recur every P = n_local / T
        // tasks; group g owns the tasks g, g + G, g + 2G, ...: relative to the image its tasks DRIFT by d = (m P mod G, to the nearer
        // multiple) per m samples, and unless that drift carries the group across the whole image within the batch's samples
        // (d x samples / m >= P) a group sees the same part of the image in every sample, and the groups that own the mesh's
        // pixels finish last.  251 is prime and drifts enough for the usual films -- but one rank's 536 rows of a 1920-wide film at
        // T = 4096 have P = 251.25: d = 0.25, 128 tasks in 512 samples, half the image; that share ran 16 % slower than its 544-row
        // sibling (58.2 against 50.3 ms: tests/tools/stripe_scaling.py c4).  So: 251 unless it resonates, then the next prime below
        // that does not.
        unsigned task_groups = HRT_TASK_GROUPS;
        {
            const double P = (double)n_local / (double)T;
            auto covers = [&](unsigned G) {       // the smallest "fraction of the image a group gets to see", over m = 1..4
                double worst = 1e30;
                for (int m = 1; m <= 4; ++m) {
                    const double r = std::fmod(m * P, (double)G), d = std::min(r, (double)G - r);
                    worst = std::min(worst, d * (double)c / (m * P));
                }
                return worst;
            };
            if (P > 1.0 && c > 1 && covers(HRT_TASK_GROUPS) < 1.0) {
                double best = -1.0;
                for (unsigned G : {251u, 241u, 239u, 233u, 229u, 227u, 223u, 211u, 199u, 197u, 193u, 191u}) {
                    const double v = covers(G);
                    if (v >= 1.0) { task_groups = G; break; }
                    if (v > best) { best = v; task_groups = G; }
                }
            }
            if (const char* e = getenv("HRT_WF_TASK_GROUPS")) task_groups = (unsigned)std::min(256, std::max(1, atoi(e)));   // experiments
        }
        // every launch of the batch gets its own zeroed block of "next task" counters
        unsigned* const ctr_base = sc->wf.buf.task_ctr;
        HIPCHK(hipMemsetAsync(ctr_base, 0, (wf_counter_words(D, n_mesh) + wf_ref_counter_words(D, n_mesh)) * sizeof(unsigned), stream));
        // ref-walk list counters: block (r, m) belongs to the traversal launch of round r, mesh m; the last block is a sink
        unsigned* const ref_base = ctr_base + wf_counter_words(D, n_mesh);
        const int nm1 = std::max(1, n_mesh);
        auto ref_block = [&](int r, int m) { return ref_base + 512 * (size_t)((r < D && m < nm1) ? r * nm1 + m : D * nm1); };
        size_t launch_no = 0;
        const bool force_dynamic = getenv("HRT_WF_DYNAMIC_TASKS") != nullptr;   // tests: exercise the pull path on small tiles too
        auto next_counters = [&](unsigned waves) {
            w.task_ctr = ctr_base + 256 * launch_no++;
            w.n_groups = std::min<unsigned>(task_groups, waves);
            w.pull_k = (w.n_tasks <= waves && !force_dynamic) ? 0u : std::min(32u, std::max(1u, w.n_tasks / (waves * 4u)));
            w.group_q = w.n_tasks / w.n_groups; w.group_r = w.n_tasks % w.n_groups;
        };
        next_counters((unsigned)task_blocks * 4u);
        w.ref_prod = ref_block(0, 0); w.ref_cons = ref_block(D, 0);
        if (stats) hipLaunchKernelGGL(k_wf_gen<true>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *cam, *pr, map, ws, n_local, s0, n_slots, w, sc->d_counters);
        else hipLaunchKernelGGL(k_wf_gen<false>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *cam, *pr, map, ws, n_local, s0, n_slots, w, sc->d_counters);
        for (int r = 0; r < tail_round; ++r) {
            const int par = r & 1;
            for (int m = 0; m < n_mesh; ++m) {
                const int mp = sc->mesh_prims[m];
                if (m > 0) {   // further meshes: analytic prims between the meshes + preparation
                    const int p0 = sc->mesh_prims[m - 1] + 1;
                    next_counters((unsigned)task_blocks * 4u);
                    w.ref_prod = ref_block(r, m); w.ref_cons = ref_block(D, 0);
                    if (stats) hipLaunchKernelGGL(k_wf_pre<true>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *pr, map, n_local, s0, r, par, p0, mp, w, sc->d_counters);
                    else hipLaunchKernelGGL(k_wf_pre<false>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *pr, map, n_local, s0, r, par, p0, mp, w, sc->d_counters);
                }
                hipEvent_t ea = nullptr, eb = nullptr;
                if (timing) { st = get_event(sc, &ea); if (st != HRT_OK) return st; st = get_event(sc, &eb); if (st != HRT_OK) return st; HIPCHK(hipEventRecord(ea, stream)); }
                // stack depth variant: LDS per block = DEPTH KB; resident blocks per CU: LDS 160 KB and 78 VGPRs -> 6 waves/SIMD
                const int md = sc->mesh_depths[m];
                const int variant = md <= 20 ? 20 : (md <= 24 ? 24 : 32);
                const int ext_blocks = sc->n_cus * (ext_per_cu_env ? ext_per_cu_env : (variant == 32 ? 4 : 6));
                next_counters((unsigned)ext_blocks * (HRT_BLOCK / 64));
                w.ref_prod = ref_block(D, 0); w.ref_cons = ref_block(r, m);
#define HRT_LAUNCH_EXT(S, D) hipLaunchKernelGGL((k_wf_ext<S, D>), dim3(ext_blocks), dim3(HRT_BLOCK), 0, stream, sc->ds, *pr, mp, par, w, sc->d_counters, leaf_num)
                if (stats) { if (variant == 20) HRT_LAUNCH_EXT(true, 20); else if (variant == 24) HRT_LAUNCH_EXT(true, 24); else HRT_LAUNCH_EXT(true, 32); }
                else { if (variant == 20) HRT_LAUNCH_EXT(false, 20); else if (variant == 24) HRT_LAUNCH_EXT(false, 24); else HRT_LAUNCH_EXT(false, 32); }
#undef HRT_LAUNCH_EXT
                if (timing) { HIPCHK(hipEventRecord(eb, stream)); sc->pending_trav.push_back({ea, eb}); }
            }
            if (stale_ff) {
                next_counters((unsigned)task_blocks * 4u);
                hipLaunchKernelGGL(k_wf_stale, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *pr, par, w);
            }
            next_counters((unsigned)task_blocks * 4u);
            w.ref_prod = ref_block(r + 1, 0); w.ref_cons = ref_block(D, 0);
            if (stats) hipLaunchKernelGGL(k_wf_shade<true>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *pr, map, ws, n_local, s0, r, w, sc->d_counters);
            else hipLaunchKernelGGL(k_wf_shade<false>, dim3(task_blocks), dim3(256), 0, stream, sc->ds, *pr, map, ws, n_local, s0, r, w, sc->d_counters);
            if (progress)   // paths ended so far = earlier batches + this batch's slots - the live ones (w.live, as k_wf_shade left it)
                hipLaunchKernelGGL(k_wf_progress, dim3(1), dim3(256), 0, stream, w.live, w.n_tasks, sc->progress_base + n_slots, sc->d_progress);
        }
        if (tail_round < D) {   // the remaining rounds of every task by the wave that pulls it
            TailMeshes tm{};
            tm.n = n_mesh;
            int need = 0;
            for (int m = 0; m < n_mesh; ++m) { tm.prim[m] = sc->mesh_prims[m]; need = std::max(need, sc->mesh_depths[m]); }
            const int variant = need <= 20 ? 20 : (need <= 24 ? 24 : 32);
            const int tail_blocks = (int)std::min<size_t>(((size_t)w.n_tasks + 3) / 4, (size_t)sc->n_cus * (variant == 32 ? 3 : 4));   // 38..50 KB of LDS per block
            next_counters((unsigned)tail_blocks * 4u);
            w.ref_prod = ref_block(D, 0); w.ref_cons = ref_block(D, 0);
#define HRT_LAUNCH_TAIL(S, DP) hipLaunchKernelGGL((k_wf_tail<S, DP>), dim3(tail_blocks), dim3(256), 0, stream, sc->ds, *pr, map, ws, tm, n_local, s0, tail_round, D, w, sc->d_counters, leaf_num)
            if (stats) { if (variant == 20) HRT_LAUNCH_TAIL(true, 20); else if (variant == 24) HRT_LAUNCH_TAIL(true, 24); else HRT_LAUNCH_TAIL(true, 32); }
            else { if (variant == 20) HRT_LAUNCH_TAIL(false, 20); else if (variant == 24) HRT_LAUNCH_TAIL(false, 24); else HRT_LAUNCH_TAIL(false, 32); }
#undef HRT_LAUNCH_TAIL
        }
        const int rblocks = (int)std::min<size_t>((n_local + 255) / 256, (size_t)sc->n_cus * 8);
        hipLaunchKernelGGL(k_wf_reduce, dim3(rblocks), dim3(256), 0, stream, w.rad, n_local, c, s0 == 0 ? 1 : 0, s0 + c >= pr->samples ? 1 : 0, pr->samples, d_out, sc->d_counters, w.wave_rays, w.n_wave_rays);
        if (progress) {
            sc->progress_base += n_slots;
            hipLaunchKernelGGL(k_set_progress, dim3(1), dim3(1), 0, stream, sc->progress_base, sc->d_progress);
        }
        HIPCHK(hipGetLastError());
    }
    return HRT_OK;
}

hrt_status launch_pathtrace(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, const RenderMap& map, float* d_out,
                            hipStream_t stream, int s_first = 0, int s_count = -1) {
    if (map.total_items <= 0) return HRT_OK;
    if (s_count < 0) s_count = pr->samples - s_first;
    if (s_first < 0 || s_count < 1 || s_first + s_count > pr->samples) return fail(HRT_ERR_INVALID, "sample range outside [0, samples)");
    if ((pr->flags & HRT_FLAG_MEGAKERNEL) && (s_first != 0 || s_count != pr->samples))
        return fail(HRT_ERR_UNSUPPORTED, "the megakernel path renders all samples in one launch (no partial sample ranges)");
    if (pr->flags & HRT_FLAG_PROGRESS) {
        // (reset in stream order: an earlier asynchronous call's last write may still be on its way)
        hipLaunchKernelGGL(k_set_progress, dim3(1), dim3(1), 0, stream, 0ull, sc->d_progress);
        sc->progress_base = 0;
        sc->progress_total = (unsigned long long)map.rw * (unsigned long long)map.rh * (unsigned long long)s_count;
    }
    hipEvent_t a, b;
    hrt_status st = get_event(sc, &a);
    if (st != HRT_OK) return st;
    st = get_event(sc, &b);
    if (st != HRT_OK) return st;
    HIPCHK(hipEventRecord(a, stream));
    st = (pr->flags & HRT_FLAG_MEGAKERNEL) ? launch_megakernel(sc, cam, pr, map, d_out, stream)
                                           : launch_wavefront(sc, cam, pr, map, d_out, stream, s_first, s_count);
    if (st != HRT_OK) return st;
    if (pr->flags & HRT_FLAG_PROGRESS) {   // whatever path rendered: everything has ended
        hipLaunchKernelGGL(k_set_progress, dim3(1), dim3(1), 0, stream, (unsigned long long)sc->progress_total, sc->d_progress);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(b, stream));
    sc->pending.push_back({a, b});
    sc->launches++;
    return HRT_OK;
}

hrt_status fold_pending(hrt_scene* sc) {
    for (auto& p : sc->pending) {
        HIPCHK(hipEventSynchronize(p.b));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        sc->kernel_ms += ms;
        sc->event_pool.push_back(p.a); sc->event_pool.push_back(p.b);
    }
    sc->pending.clear();
    for (auto& p : sc->pending_trav) {
        HIPCHK(hipEventSynchronize(p.b));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        sc->traversal_ms += ms;
        sc->traversal_launches++;
        sc->event_pool.push_back(p.a); sc->event_pool.push_back(p.b);
    }
    sc->pending_trav.clear();
    return HRT_OK;
}

}  // namespace

// No C++ exception may cross the C ABI (std::bad_alloc from the packers' vectors, std::length_error ...).
#define HRT_API_TRY try {
#define HRT_API_CATCH \
    } catch (const std::bad_alloc&) { return fail(HRT_ERR_OOM, "out of host memory"); } \
    catch (const std::exception& e) { return fail(HRT_ERR_INVALID, e.what()); } \
    catch (...) { return fail(HRT_ERR_INVALID, "unknown C++ exception"); }

extern "C" {

const char* hrt_status_str(hrt_status s) {
    switch (s) {
        case HRT_OK: return "ok";
        case HRT_ERR_INVALID: return "invalid argument";
        case HRT_ERR_HIP: return "HIP runtime error";
        case HRT_ERR_NO_DEVICE: return "no GPU device";
        case HRT_ERR_OOM: return "out of device memory";
        case HRT_ERR_IO: return "I/O error";
        case HRT_ERR_PARSE: return "parse error";
        case HRT_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown";
}
const char* hrt_last_error(void) { return g_err.c_str(); }

hrt_status hrt_debug_bounds_violations(int32_t device, int64_t* out8) {
    HRT_API_TRY
    if (!out8) return fail(HRT_ERR_INVALID, "NULL argument");
#ifdef HRT_DEBUG_BOUNDS
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long v[HRT_BOUNDS_SLOTS];
    HIPCHK(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_hrt_bounds_violations), sizeof(v)));
    const unsigned long long zero[HRT_BOUNDS_SLOTS] = {};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_hrt_bounds_violations), zero, sizeof(zero)));
    for (int k = 0; k < 8; ++k) out8[k] = k < HRT_BOUNDS_SLOTS ? (int64_t)v[k] : 0;
    return HRT_OK;
#else
    (void)device;
    return fail(HRT_ERR_UNSUPPORTED, "this library was built without -DHRT_DEBUG_BOUNDS");
#endif
    HRT_API_CATCH
}
const char* hrt_version(void) { return "hrt-mi355x 0.1 (gfx950)"; }

hrt_status hrt_device_count(int* n) {
    HRT_API_TRY
    if (!n) return fail(HRT_ERR_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail_hip(e, "hipGetDeviceCount"); }
    *n = c;
    return HRT_OK;
    HRT_API_CATCH
}

void hrt_scene_destroy(hrt_scene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    for (auto& p : sc->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : sc->pending_trav) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : sc->event_pool) (void)hipEventDestroy(e);
    if (sc->wf.base) (void)hipFree(sc->wf.base);
    for (void* p : sc->allocs) (void)hipFree(p);
    if (sc->h_progress) (void)hipHostFree((void*)sc->h_progress);
    delete sc;
}

hrt_status hrt_scene_create(const hrt_flat_scene* f, int device, hrt_scene** out) {
    HRT_API_TRY
    if (!out) return fail(HRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    hrt_status st = validate(f);
    if (st != HRT_OK) return st;
    // the reference's own tree of every mesh, for the rays that must walk it (hrt_device.h ref_walk); part of the validation:
    // tri_ref_order must be the depth-first code of such a tree (checked before any device is touched)
    RefTree ref;
    {
        std::string why;
        if (!pack_ref_tree(f, ref, &why)) return fail(HRT_ERR_INVALID, why);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(HRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(HRT_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));

    hrt_scene* sc = new (std::nothrow) hrt_scene;
    if (!sc) return fail(HRT_ERR_OOM, "host allocation failed");
    sc->device = device;
    sc->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

#define UP(dst, src, bytes)                                              \
    do {                                                                 \
        st = upload(&(dst), (src), (bytes));                             \
        if (st != HRT_OK) { hrt_scene_destroy(sc); return st; }          \
        sc->allocs.push_back((void*)(dst));                              \
    } while (0)

    hrt_prim* d_prims; hrt_material* d_mats; hrt_texture* d_texs; hrt_mesh* d_meshes;
    UP(d_prims, f->prims, sizeof(hrt_prim) * f->n_prims);
    UP(d_mats, f->materials, sizeof(hrt_material) * f->n_materials);
    UP(d_texs, f->textures, sizeof(hrt_texture) * f->n_textures);
    UP(d_meshes, f->meshes, sizeof(hrt_mesh) * f->n_meshes);
    // repack the BVH into 32-byte culling records and the triangles into 16-byte aligned records (hrt_pack.h)
    std::vector<uint32_t> qn;
    std::vector<float> grids;
    int node_layout = 0;                                  // experiments: HRT_NODE_LAYOUT=treelet (hrt_pack.h treelet_order)
    if (const char* e = getenv("HRT_NODE_LAYOUT")) node_layout = strcmp(e, "treelet") == 0 ? 1 : 0;
    pack_nodes(f, qn, grids, node_layout);
    uint4* d_nodes; float4* d_grids;
    UP(d_nodes, qn.data(), qn.size() * sizeof(uint32_t));
    UP(d_grids, grids.data(), grids.size() * sizeof(float));
    uint4 *d_rnodes, *d_rmesh; float4* d_rtris;
    UP(d_rnodes, ref.nodes.data(), ref.nodes.size() * sizeof(uint32_t));
    UP(d_rtris, ref.tris.data(), ref.tris.size() * sizeof(float));
    UP(d_rmesh, ref.mesh_nodes.data(), ref.mesh_nodes.size() * sizeof(uint32_t));
    std::vector<float> pos, attr, box;
    pack_triangles(f, ref, pos, attr, box);
    float4 *d_pos, *d_attr, *d_box;
    UP(d_pos, pos.data(), pos.size() * sizeof(float));
    UP(d_attr, attr.data(), attr.size() * sizeof(float));
    UP(d_box, box.data(), box.size() * sizeof(float));
    uint8_t* d_u8; float* d_f32;
    UP(d_u8, f->texels_u8, (size_t)f->n_texels_u8);
    UP(d_f32, f->texels_f32, (size_t)f->n_texels_f32 * sizeof(float));
    DeviceCounters zero{};
    UP(sc->d_counters, &zero, sizeof(zero));
    unsigned zw = 0;
    UP(sc->d_work, &zw, sizeof(zw));
#undef UP
    {   // the progress word: host memory the device can write and any host thread read without a HIP call (hrt_scene_progress);
        // made here, not at the first render, so that a reporter thread never sees the pointer change
        void* h = nullptr;
        e = hipHostMalloc(&h, 64, hipHostMallocMapped);
        if (e != hipSuccess) { hrt_scene_destroy(sc); return fail_hip(e, "hipHostMalloc (progress counter)"); }
        memset(h, 0, 64);
        void* d = nullptr;
        e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) { (void)hipHostFree(h); hrt_scene_destroy(sc); return fail_hip(e, "hipHostGetDevicePointer"); }
        sc->h_progress = (volatile unsigned long long*)h; sc->d_progress = (unsigned long long*)d;
    }

    sc->ds.prims = d_prims; sc->ds.mats = d_mats; sc->ds.texs = d_texs; sc->ds.meshes = d_meshes;
    sc->ds.qnodes = d_nodes; sc->ds.grids = d_grids; sc->ds.tri_pos = d_pos; sc->ds.tri_attr = d_attr; sc->ds.tri_box = d_box;
    sc->ds.rnodes = d_rnodes; sc->ds.rtris = d_rtris; sc->ds.rmesh = d_rmesh;
    {   // experiments: the shear amplification |d| / |d[kZ]| from which a Q-4 ray walks the reference's tree
        float a = HRT_Q4_ROUTE_A_DEFAULT;
        if (const char* e = getenv("HRT_Q4_ROUTE_A")) { const float v = (float)atof(e); if (v > 0.0f) a = v; }
        sc->ds.q4_route_a2 = a * a;
        sc->ds.ref_fold_all = getenv("HRT_REF_FOLD_ALL") ? 1 : 0;   // tests: the megakernel / test kernels fold instead of walking
        sc->ds.stale_ff = scene_has_stale_front_face(f);
    }
    sc->ds.texels_u8 = d_u8; sc->ds.texels_f32 = d_f32;
    sc->ds.n_prims = (int32_t)f->n_prims;
    sc->ds.background_tex = f->background_tex;
    sc->ds.lprims = d_prims; sc->ds.lmats = d_mats; sc->ds.ltexs = d_texs; sc->ds.lmeshes = d_meshes;
    sc->ds.n_mats = (int32_t)f->n_materials; sc->ds.n_texs = (int32_t)f->n_textures; sc->ds.n_meshes = (int32_t)f->n_meshes;
    sc->n_prims = (int)f->n_prims;
    for (uint32_t i = 0; i < f->n_prims; ++i)
        if (f->prims[i].kind == HRT_PRIM_MESH) {
            sc->mesh_prims.push_back((int)i);
            sc->mesh_depths.push_back(bvh_depth(f, f->meshes[f->prims[i].mesh]));
        }
    *out = sc;
    return HRT_OK;
    HRT_API_CATCH
}

int32_t hrt_stripe_rows(int32_t height, int32_t R, int32_t rank, int32_t G) {
    if (height <= 0 || R <= 0 || G <= 0 || rank < 0 || rank >= G) return 0;
    const int32_t n_blocks = (height + R - 1) / R;
    int32_t rows = 0;
    for (int32_t b = rank; b < n_blocks; b += G) {
        const int32_t r0 = b * R;
        rows += (r0 + R <= height) ? R : (height - r0);
    }
    return rows;
}
int32_t hrt_stripe_row_index(int32_t height, int32_t R, int32_t rank, int32_t G, int32_t local) {
    if (local < 0 || local >= hrt_stripe_rows(height, R, rank, G)) return -1;
    const int32_t b = local / R;
    return (b * G + rank) * R + (local - b * R);
}

hrt_status hrt_render_stripes_device(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, int32_t R, int32_t rank,
                                     int32_t G, float* d_out, void* stream) {
    HRT_API_TRY
    if (!sc || !cam || !d_out) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    if (R <= 0 || G <= 0 || rank < 0 || rank >= G) return fail(HRT_ERR_INVALID, "bad stripe partition");
    HIPCHK(hipSetDevice(sc->device));
    RenderMap map{};
    map.mode = 1; map.R = R; map.rank = rank; map.G = G;
    map.rw = pr->width; map.rh = hrt_stripe_rows(pr->height, R, rank, G);
    map.tiles_x = (map.rw + 7) / 8;
    map.total_items = map.tiles_x * ((map.rh + 7) / 8) * 64;
    map.m_rw = fastdiv_magic((uint32_t)map.rw); map.m_nl = fastdiv_magic((uint32_t)map.rw * (uint32_t)map.rh);
    return launch_pathtrace(sc, cam, pr, map, d_out, (hipStream_t)stream);
    HRT_API_CATCH
}

hrt_status hrt_render_stripes_accumulate_device(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, int32_t R, int32_t rank,
                                                int32_t G, float* d_accum, int32_t sample_first, int32_t sample_count, void* stream) {
    HRT_API_TRY
    if (!sc || !cam || !d_accum) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    if (R <= 0 || G <= 0 || rank < 0 || rank >= G) return fail(HRT_ERR_INVALID, "bad stripe partition");
    HIPCHK(hipSetDevice(sc->device));
    RenderMap map{};
    map.mode = 1; map.R = R; map.rank = rank; map.G = G;
    map.rw = pr->width; map.rh = hrt_stripe_rows(pr->height, R, rank, G);
    map.tiles_x = (map.rw + 7) / 8;
    map.total_items = map.tiles_x * ((map.rh + 7) / 8) * 64;
    map.m_rw = fastdiv_magic((uint32_t)map.rw); map.m_nl = fastdiv_magic((uint32_t)map.rw * (uint32_t)map.rh);
    return launch_pathtrace(sc, cam, pr, map, d_accum, (hipStream_t)stream, sample_first, sample_count);
    HRT_API_CATCH
}

hrt_status hrt_render_stripes_accumulate(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, int32_t R, int32_t rank, int32_t G,
                                         float* accum, int32_t sample_first, int32_t sample_count, hrt_stats* stats) {
    HRT_API_TRY
    if (!sc || !cam || !accum) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    if (R <= 0 || G <= 0 || rank < 0 || rank >= G) return fail(HRT_ERR_INVALID, "bad stripe partition");
    HIPCHK(hipSetDevice(sc->device));
    hrt_stats prev;
    st = hrt_scene_stats(sc, &prev);
    if (st != HRT_OK) return st;
    const size_t bytes = (size_t)hrt_stripe_rows(pr->height, R, rank, G) * pr->width * 3 * sizeof(float);
    if (bytes) {
        float* d = nullptr;
        hipError_t e = hipMalloc((void**)&d, bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        if (sample_first > 0) {
            hipError_t e1 = hipMemcpy(d, accum, bytes, hipMemcpyHostToDevice);
            if (e1 != hipSuccess) { (void)hipFree(d); return fail_hip(e1, "hipMemcpy H2D accumulation buffer"); }
        }
        st = hrt_render_stripes_accumulate_device(sc, cam, pr, R, rank, G, d, sample_first, sample_count, nullptr);
        if (st == HRT_OK) {
            hipError_t e2 = hipMemcpy(accum, d, bytes, hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) st = fail_hip(e2, "hipMemcpy D2H accumulation buffer");
        }
        (void)hipFree(d);
        if (st != HRT_OK) return st;
    }
    hrt_stats now;
    st = hrt_scene_stats(sc, &now);
    if (st != HRT_OK) return st;
    if (stats) *stats = now;
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_scene_stats(hrt_scene* sc, hrt_stats* stats) {
    HRT_API_TRY
    if (!sc || !stats) return fail(HRT_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(sc->device));
    HIPCHK(hipDeviceSynchronize());
    hrt_status st = fold_pending(sc);
    if (st != HRT_OK) return st;
    DeviceCounters c;
    HIPCHK(hipMemcpy(&c, sc->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(sc->d_counters, 0, sizeof(c)));
#ifdef HRT_STEP_PROFILE
    if (c.prof[5] && c.prof[6]) {
        const double n = (double)c.prof[5], a = (double)c.prof[6], step = c.prof[4] / n, l = c.prof[0] / a, b = c.prof[1] / a, d = c.prof[2] / a;
        fprintf(stderr, "[step profile] %.4g node steps of a wave in k_wf_ext, lane 0 taking part in %.0f %%; s_memtime ticks per step %.0f = node loads until the data is there %.0f | box tests %.0f | "
                        "descend / push / pop until the LDS answers %.0f | votes and loop control %.0f  (each part includes one s_memtime round trip)\n",
                n, 100.0 * a / n, step, l, b, d, step - l - b - d);
    }
#endif
#ifdef HRT_SHADE_PROFILE
    if (c.prof[0]) {
        double tot = 0; for (int k = 0; k < 6; ++k) tot += (double)c.prof[k];
        fprintf(stderr, "[shade profile] wave-cycles: load+trailing prims %.1f %% | miss queue %.1f %% | hitRecord+scatter %.1f %% | next-segment prepare %.1f %% | stores+enqueue %.1f %% (total %.3g)\n",
                100.0 * c.prof[0] / tot, 100.0 * c.prof[1] / tot, 100.0 * c.prof[2] / tot, 100.0 * c.prof[3] / tot, 100.0 * c.prof[4] / tot, tot);
    }
#endif
#ifdef HRT_EXT_PROFILE
    if (c.prof[0]) {
        const double in_steps = (double)c.prof[1], lf = (double)c.prof[3];
        fprintf(stderr, "[ext profile] outer %llu | inner wave-steps %llu, lanes at inner %.1f %% of 64, lanes with a ray %.1f %% | leaf phases %llu, lanes at leaf %.1f %%, "
                        "of them with 2 triangles %.1f %%, lanes with a ray %.1f %% | lanes without a ray at the end of an outer iteration %.1f %%\n",
                c.prof[0], c.prof[1], 100.0 * c.prof[2] / (64.0 * in_steps), 100.0 * c.prof[8] / (64.0 * in_steps), c.prof[3], 100.0 * c.prof[4] / (64.0 * lf),
                100.0 * c.prof[5] / (double)c.prof[4], 100.0 * c.prof[6] / (64.0 * lf), 100.0 * c.prof[7] / (64.0 * (double)c.prof[0]));
    }
#endif
    stats->rays = c.rays; stats->samples = c.samples; stats->box_tests = c.box_tests; stats->tri_tests = c.tri_tests;
    stats->mesh_hits = c.mesh_hits; stats->env_lookups = c.env_lookups;
    stats->traversal_box_tests = c.trav_box_tests; stats->traversal_tri_tests = c.trav_tri_tests;
    stats->kernel_ms = sc->kernel_ms; stats->launches = sc->launches;
    stats->traversal_ms = sc->traversal_ms; stats->traversal_launches = sc->traversal_launches;
    sc->kernel_ms = 0.0; sc->launches = 0; sc->traversal_ms = 0.0; sc->traversal_launches = 0;
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_render_tile(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, hrt_rect tile, float* out,
                           hrt_stats* stats) {
    HRT_API_TRY
    if (!sc || !cam || !out) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    if (tile.w <= 0 || tile.h <= 0 || tile.x0 < 0 || tile.y0 < 0 || tile.x0 + tile.w > pr->width || tile.y0 + tile.h > pr->height)
        return fail(HRT_ERR_INVALID, "tile outside the film");
    HIPCHK(hipSetDevice(sc->device));
    // discard counters of earlier async launches so `stats` describes this call only
    hrt_stats prev;
    st = hrt_scene_stats(sc, &prev);
    if (st != HRT_OK) return st;
    const size_t bytes = (size_t)tile.w * tile.h * 3 * sizeof(float);
    float* d_out = nullptr;
    hipError_t e = hipMalloc((void**)&d_out, bytes);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    RenderMap map{};
    map.mode = 0; map.x0 = tile.x0; map.y0 = tile.y0; map.rw = tile.w; map.rh = tile.h; map.R = 1; map.G = 1;
    map.tiles_x = (map.rw + 7) / 8;
    map.total_items = map.tiles_x * ((map.rh + 7) / 8) * 64;
    map.m_rw = fastdiv_magic((uint32_t)map.rw); map.m_nl = fastdiv_magic((uint32_t)map.rw * (uint32_t)map.rh);
    st = launch_pathtrace(sc, cam, pr, map, d_out, nullptr);
    if (st == HRT_OK) {
        e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail_hip(e, "hipMemcpy D2H film tile");
    }
    (void)hipFree(d_out);
    if (st != HRT_OK) return st;
    hrt_stats now;
    st = hrt_scene_stats(sc, &now);
    if (st != HRT_OK) return st;
    if (stats) *stats = now;
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_render_stripes(hrt_scene* sc, const hrt_camera* cam, const hrt_params* pr, int32_t R, int32_t rank, int32_t G,
                              float* out, hrt_stats* stats) {
    HRT_API_TRY
    if (!sc || !cam || !out) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    HIPCHK(hipSetDevice(sc->device));
    hrt_stats prev;
    st = hrt_scene_stats(sc, &prev);
    if (st != HRT_OK) return st;
    const int32_t rows = hrt_stripe_rows(pr->height, R, rank, G);
    if (R <= 0 || G <= 0 || rank < 0 || rank >= G) return fail(HRT_ERR_INVALID, "bad stripe partition");
    const size_t bytes = (size_t)rows * pr->width * 3 * sizeof(float);
    float* d_out = nullptr;
    if (bytes) {
        hipError_t e = hipMalloc((void**)&d_out, bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        st = hrt_render_stripes_device(sc, cam, pr, R, rank, G, d_out, nullptr);
        if (st == HRT_OK) {
            hipError_t e2 = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) st = fail_hip(e2, "hipMemcpy D2H film stripes");
        }
        (void)hipFree(d_out);
        if (st != HRT_OK) return st;
    }
    hrt_stats now;
    st = hrt_scene_stats(sc, &now);
    if (st != HRT_OK) return st;
    if (stats) *stats = now;
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_resolve_u8_device(hrt_scene* sc, const float* d_rgb, int64_t n_pixels, uint8_t* d_out, void* stream) {
    HRT_API_TRY
    if (!sc || !d_rgb || !d_out || n_pixels < 0) return fail(HRT_ERR_INVALID, "bad argument");
    if (n_pixels == 0) return HRT_OK;
    HIPCHK(hipSetDevice(sc->device));
    int blocks = (int)((n_pixels + 255) / 256);
    if (blocks > sc->n_cus * 8) blocks = sc->n_cus * 8;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_rgb, (long long)n_pixels, d_out);
    HIPCHK(hipGetLastError());
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_resolve_u8(hrt_scene* sc, const float* rgb, int64_t n_pixels, uint8_t* out) {
    HRT_API_TRY
    if (!sc || !rgb || !out || n_pixels < 0) return fail(HRT_ERR_INVALID, "bad argument");
    if (n_pixels == 0) return HRT_OK;
    HIPCHK(hipSetDevice(sc->device));
    float* d_in = nullptr; uint8_t* d_out = nullptr;
    HIPCHK(hipMalloc((void**)&d_in, (size_t)n_pixels * 12));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n_pixels * 3);
    if (e != hipSuccess) { (void)hipFree(d_in); return fail_hip(e, "hipMalloc"); }
    hrt_status st = HRT_OK;
    e = hipMemcpy(d_in, rgb, (size_t)n_pixels * 12, hipMemcpyHostToDevice);
    if (e != hipSuccess) st = fail_hip(e, "hipMemcpy H2D");
    if (st == HRT_OK) st = hrt_resolve_u8_device(sc, d_in, n_pixels, d_out, nullptr);
    if (st == HRT_OK) { e = hipMemcpy(out, d_out, (size_t)n_pixels * 3, hipMemcpyDeviceToHost); if (e != hipSuccess) st = fail_hip(e, "hipMemcpy D2H"); }
    (void)hipFree(d_in); (void)hipFree(d_out);
    return st;
    HRT_API_CATCH
}

hrt_status hrt_closest_hit(hrt_scene* sc, const hrt_params* pr, int64_t n, const float* o, const float* d, float t_min,
                           float t_max, uint32_t pixel0, hrt_hit* out) {
    HRT_API_TRY
    if (!sc || !pr || !o || !d || !out || n < 0) return fail(HRT_ERR_INVALID, "bad argument");
    if (n == 0) return HRT_OK;
    HIPCHK(hipSetDevice(sc->device));
    float *d_o = nullptr, *d_d = nullptr; hrt_hit* d_h = nullptr;
    hrt_status st = HRT_OK;
    hipError_t e;
    if ((e = hipMalloc((void**)&d_o, (size_t)n * 12)) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMalloc((void**)&d_d, (size_t)n * 12)) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMalloc((void**)&d_h, (size_t)n * sizeof(hrt_hit))) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMemcpy(d_o, o, (size_t)n * 12, hipMemcpyHostToDevice)) != hipSuccess) st = fail_hip(e, "hipMemcpy");
    if (st == HRT_OK && (e = hipMemcpy(d_d, d, (size_t)n * 12, hipMemcpyHostToDevice)) != hipSuccess) st = fail_hip(e, "hipMemcpy");
    if (st == HRT_OK) {
        const int blocks = (int)((n + HRT_BLOCK - 1) / HRT_BLOCK);
        hipLaunchKernelGGL(k_closest_hit, dim3(blocks), dim3(HRT_BLOCK), 0, 0, sc->ds, *pr, (long long)n, d_o, d_d, t_min, t_max, pixel0, d_h);
        if ((e = hipGetLastError()) != hipSuccess) st = fail_hip(e, "k_closest_hit launch");
    }
    if (st == HRT_OK && (e = hipMemcpy(out, d_h, (size_t)n * sizeof(hrt_hit), hipMemcpyDeviceToHost)) != hipSuccess) st = fail_hip(e, "hipMemcpy D2H");
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_h);
    return st;
    HRT_API_CATCH
}

hrt_status hrt_math_probe(int device, int32_t op, int64_t n, const float* in, const float* in2, float* out) {
    HRT_API_TRY
    if (!in || !out || n < 0 || op < 0 || op > 11) return fail(HRT_ERR_INVALID, "bad argument");
    if ((op == 3 || op == 5 || op == 6 || op == 7 || op == 9 || op == 10) && !in2) return fail(HRT_ERR_INVALID, "in2 required");
    if (n == 0) return HRT_OK;
    HIPCHK(hipSetDevice(device));
    const size_t n_in = (size_t)n * (op == 5 ? 4 : 1), n_in2 = (size_t)n * (op == 5 ? 2 : 1), n_out = (size_t)n * (op == 5 ? 4 : 1);
    float *d_in = nullptr, *d_in2 = nullptr, *d_out = nullptr;
    hrt_status st = HRT_OK; hipError_t e;
    if ((e = hipMalloc((void**)&d_in, n_in * 4)) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMalloc((void**)&d_in2, n_in2 * 4)) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMalloc((void**)&d_out, n_out * 4)) != hipSuccess) st = fail_hip(e, "hipMalloc");
    if (st == HRT_OK && (e = hipMemcpy(d_in, in, n_in * 4, hipMemcpyHostToDevice)) != hipSuccess) st = fail_hip(e, "hipMemcpy");
    if (st == HRT_OK && in2 && (e = hipMemcpy(d_in2, in2, n_in2 * 4, hipMemcpyHostToDevice)) != hipSuccess) st = fail_hip(e, "hipMemcpy");
    if (st == HRT_OK) {
        hipLaunchKernelGGL(k_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (long long)n, d_in, d_in2, d_out);
        if ((e = hipGetLastError()) != hipSuccess) st = fail_hip(e, "k_math_probe launch");
    }
    if (st == HRT_OK && (e = hipMemcpy(out, d_out, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) st = fail_hip(e, "hipMemcpy D2H");
    (void)hipFree(d_in); (void)hipFree(d_in2); (void)hipFree(d_out);
    return st;
    HRT_API_CATCH
}


// ---------------------------------------------------------------- multi-GPU session: scene replicated on the
// devices of this process, image rows dealt in interleaved blocks, one RCCL gather of the device-resident stripes over xGMI
struct hrt_multi {
    std::vector<int> devices;
    std::vector<hrt_scene*> scenes;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;       // empty until the first gather that needs them
    std::vector<float*> d_accum;         // per rank: `share` floats (its stripes, padded to the largest share)
    std::vector<float*> d_gather;        // per rank: G x share floats (ncclAllGather's receive buffer)
    float* d_film = nullptr;             // first device: H x W x 3 sums in film order
    float* d_mean = nullptr;             // first device: preview means
    uint8_t* d_u8 = nullptr;
    int W = 0, H = 0, R = 0;
    long long share = 0;
    bool use_rccl = false;
    bool loopback = false;               // test mode (force_rccl < 0): logical ranks may share a device, the gather is G device copies
    bool poisoned = false;               // a rank failed inside a render: the ranks' sums no longer describe the same sample range
};

namespace {
// RCCL is loaded when the first session asks for a communicator, not linked: a process that also hosts PyTorch (bench.py, the
// tests) already has PyTorch's own copy of librccl, and two copies in one process end in a double free at exit.  dlopen by
// soname hands back the copy that is already there, or /opt/rocm/lib's.
struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
}  // namespace
}  // extern "C"
namespace {
const RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.CommInitAll && a.CommDestroy && a.AllGather && a.GroupStart && a.GroupEnd && a.GetErrorString;
        return a;
    }();
    return api;
}
}  // namespace
extern "C" {
namespace {
hrt_status fail_nccl(ncclResult_t r, const char* what) { g_err = std::string(what) + ": " + (rccl_api().ok ? rccl_api().GetErrorString(r) : "RCCL not loaded"); return HRT_ERR_HIP; }
#define NCCLCHK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail_nccl(_r, #expr); } while (0)

void multi_free_buffers(hrt_multi* m) {
    for (size_t g = 0; g < m->devices.size(); ++g) {
        (void)hipSetDevice(m->devices[g]);
        if (g < m->d_gather.size() && m->d_gather[g] && (g >= m->d_accum.size() || m->d_gather[g] != m->d_accum[g])) (void)hipFree(m->d_gather[g]);
        if (g < m->d_accum.size() && m->d_accum[g]) (void)hipFree(m->d_accum[g]);
    }
    m->d_accum.clear(); m->d_gather.clear();
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->d_film) (void)hipFree(m->d_film);
    if (m->d_mean) (void)hipFree(m->d_mean);
    if (m->d_u8) (void)hipFree(m->d_u8);
    m->d_film = m->d_mean = nullptr; m->d_u8 = nullptr;
    m->W = m->H = m->R = 0; m->share = 0;
}
hrt_status multi_reserve(hrt_multi* m, int W, int H, int R) {
    if (m->W == W && m->H == H && m->R == R && !m->d_accum.empty()) return HRT_OK;
    multi_free_buffers(m);
    const int G = (int)m->devices.size();
    m->share = (long long)hrt_stripe_rows(H, R, 0, G) * W * 3;        // rank 0 owns the most rows
    if (m->share == 0) m->share = 4;
    m->d_accum.assign(G, nullptr); m->d_gather.assign(G, nullptr);
    for (int g = 0; g < G; ++g) {
        HIPCHK(hipSetDevice(m->devices[g]));
        HIPCHK(hipMalloc((void**)&m->d_accum[g], (size_t)m->share * sizeof(float)));
        HIPCHK(hipMemset(m->d_accum[g], 0, (size_t)m->share * sizeof(float)));
        if (m->use_rccl || (m->loopback && g == 0 && G > 1)) HIPCHK(hipMalloc((void**)&m->d_gather[g], (size_t)m->share * G * sizeof(float)));
    }
    if (!m->d_gather[0]) m->d_gather[0] = m->d_accum[0];               // one rank, no communicator: its stripes ARE the gathered buffer
    HIPCHK(hipSetDevice(m->devices[0]));
    HIPCHK(hipMalloc((void**)&m->d_film, (size_t)W * H * 3 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&m->d_mean, (size_t)W * H * 3 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&m->d_u8, (size_t)W * H * 3));
    m->W = W; m->H = H; m->R = R;
    return HRT_OK;
}
}  // namespace

void hrt_multi_destroy(hrt_multi* m) {
    if (!m) return;
    multi_free_buffers(m);
    for (size_t g = 0; g < m->comms.size(); ++g) if (m->comms[g]) (void)rccl_api().CommDestroy(m->comms[g]);
    for (size_t g = 0; g < m->streams.size(); ++g) { (void)hipSetDevice(m->devices[g]); if (m->streams[g]) (void)hipStreamDestroy(m->streams[g]); }
    for (hrt_scene* s : m->scenes) hrt_scene_destroy(s);
    delete m;
}

hrt_status hrt_multi_create(const hrt_flat_scene* flat, int32_t n_devices, const int32_t* devices, int32_t force_rccl, hrt_multi** out) {
    HRT_API_TRY
    if (!out) return fail(HRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_devices < 1) return fail(HRT_ERR_INVALID, "n_devices must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(HRT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    hrt_multi* m = new hrt_multi;
    for (int g = 0; g < n_devices; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= ndev) { delete m; return fail(HRT_ERR_INVALID, "device index out of range"); }
        if (force_rccl >= 0) for (int k : m->devices) if (k == d) { delete m; return fail(HRT_ERR_INVALID, "device listed twice"); }
        m->devices.push_back(d);
    }
    // force_rccl < 0: LOOPBACK, a test mode for boxes with fewer devices than ranks -- a device may be listed more than once and
    // the gather is one device-to-device copy per rank instead of ncclAllGather (RCCL refuses two ranks on one device); everything
    // else (a host thread, a stream and a scene per rank, padded shares, idle ranks, k_unstripe / k_restripe) is the production code
    m->loopback = force_rccl < 0;
    m->use_rccl = !m->loopback && (n_devices > 1 || force_rccl != 0);
    m->scenes.assign(n_devices, nullptr); m->streams.assign(n_devices, nullptr);
    for (int g = 0; g < n_devices; ++g) {
        hrt_status st = hrt_scene_create(flat, m->devices[g], &m->scenes[g]);
        if (st != HRT_OK) { hrt_multi_destroy(m); return st; }
        hipError_t e = hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking);
        if (e != hipSuccess) { hrt_multi_destroy(m); return fail_hip(e, "hipStreamCreate"); }
    }
    if (m->use_rccl) {   // one communicator per device of this process (single process: ncclCommInitAll)
        if (!rccl_api().ok) { hrt_multi_destroy(m); return fail(HRT_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded (needed for more than one device)"); }
        m->comms.assign(n_devices, nullptr);
        ncclResult_t r = rccl_api().CommInitAll(m->comms.data(), n_devices, m->devices.data());
        if (r != ncclSuccess) { m->comms.clear(); hrt_multi_destroy(m); return fail_nccl(r, "ncclCommInitAll"); }
    }
    *out = m;
    return HRT_OK;
    HRT_API_CATCH
}

hrt_status hrt_scene_progress(const hrt_scene* sc, uint64_t* done, uint64_t* total) {
    if (!sc || !done || !total) return fail(HRT_ERR_INVALID, "NULL argument");
    *done = sc->h_progress ? (uint64_t)*sc->h_progress : 0;      // (plain reads of host memory: no HIP call, any thread)
    *total = (uint64_t)sc->progress_total;
    if (*done > *total) *done = *total;
    return HRT_OK;
}
hrt_status hrt_multi_progress(const hrt_multi* m, uint64_t* done, uint64_t* total) {
    if (!m || !done || !total) return fail(HRT_ERR_INVALID, "NULL argument");
    uint64_t d = 0, t = 0;
    for (const hrt_scene* s : m->scenes) {
        uint64_t a = 0, b = 0;
        if (s && hrt_scene_progress(s, &a, &b) == HRT_OK) { d += a; t += b; }
    }
    *done = d; *total = t;
    return HRT_OK;
}

int32_t hrt_multi_devices(const hrt_multi* m) { return m ? (int32_t)m->devices.size() : 0; }
int32_t hrt_multi_uses_rccl(const hrt_multi* m) { return m && m->use_rccl ? 1 : 0; }

hrt_status hrt_multi_render(hrt_multi* m, const hrt_camera* cam, const hrt_params* pr, int32_t R, int32_t sample_first, int32_t sample_count,
                            const float* resume_sums, float* out_sums, uint8_t* out_u8, hrt_stats* stats) {
    HRT_API_TRY
    if (!m || !cam) return fail(HRT_ERR_INVALID, "NULL argument");
    hrt_status st = check_params(pr);
    if (st != HRT_OK) return st;
    if (R <= 0) return fail(HRT_ERR_INVALID, "rows_per_block must be positive");
    if (sample_count < 0) sample_count = pr->samples - sample_first;
    if (sample_first < 0 || sample_count < 0 || sample_first + sample_count > pr->samples) return fail(HRT_ERR_INVALID, "sample range outside [0, samples)");
    const int G = (int)m->devices.size();
    const int W = pr->width, H = pr->height, W3 = W * 3;
    st = multi_reserve(m, W, H, R);
    if (st != HRT_OK) return st;
    if (m->poisoned) {
        if (!resume_sums && !(sample_first == 0 && sample_count > 0))
            return fail(HRT_ERR_INVALID, "an earlier render of this session failed on one rank after others had added their samples: pass resume_sums or start again at sample 0");
        m->poisoned = false;
    }
    const long long n_film = (long long)H * W3;
    auto blocks_for = [&](long long n) { return (int)std::min<long long>((n + 255) / 256, (long long)m->scenes[0]->n_cus * 8); };

    if (resume_sums) {   // a checkpoint's sums (film order) -> every rank's stripes
        HIPCHK(hipSetDevice(m->devices[0]));
        HIPCHK(hipMemcpyAsync(m->d_film, resume_sums, (size_t)n_film * sizeof(float), hipMemcpyHostToDevice, m->streams[0]));
        HIPCHK(hipStreamSynchronize(m->streams[0]));
        for (int g = 0; g < G; ++g) {
            const int rows = hrt_stripe_rows(H, R, g, G);
            if (!rows) continue;
            HIPCHK(hipSetDevice(m->devices[g]));
            float* src = m->d_film;
            float* tmp = nullptr;
            if (g != 0) {   // (rare path: through the host rather than a second collective)
                HIPCHK(hipMalloc((void**)&tmp, (size_t)n_film * sizeof(float)));
                HIPCHK(hipMemcpy(tmp, resume_sums, (size_t)n_film * sizeof(float), hipMemcpyHostToDevice));
                src = tmp;
            }
            hipLaunchKernelGGL(k_restripe, dim3(blocks_for((long long)rows * W3)), dim3(256), 0, m->streams[g], src, m->d_accum[g], H, W3, R, G, g, rows);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(m->streams[g]));
            if (tmp) (void)hipFree(tmp);
        }
    }

    // ---- render: every device its stripes, concurrently (one host thread per device enqueues its ~100 launches)
    std::vector<hrt_status> rst(G, HRT_OK);
    std::vector<std::string> rerr(G);
    if (sample_count > 0) {
        auto work = [&](int g) {
            if (hrt_stripe_rows(H, R, g, G) == 0) return;      // more ranks than row blocks: nothing to do here
            rst[g] = hrt_render_stripes_accumulate_device(m->scenes[g], cam, pr, R, g, G, m->d_accum[g], sample_first, sample_count, m->streams[g]);
            if (rst[g] != HRT_OK) rerr[g] = g_err;
        };
        {
            struct Joiner {      // an exception (std::thread's constructor can throw) must not destroy joinable threads: std::terminate
                std::vector<std::thread> threads;
                ~Joiner() { for (auto& t : threads) if (t.joinable()) t.join(); }
            } pool;
            pool.threads.reserve(G);
            for (int g = 1; g < G; ++g) pool.threads.emplace_back(work, g);
            work(0);
        }
        for (int g = 0; g < G; ++g) if (rst[g] != HRT_OK) {
            // the other ranks HAVE added this sample range to their sums: the session's sums are inconsistent from here on
            m->poisoned = true;
            return fail(rst[g], "device " + std::to_string(m->devices[g]) + " (rank " + std::to_string(g) + "): " + rerr[g] +
                                " -- the session's sums are now inconsistent across ranks: continue with resume_sums (a checkpoint) or from sample 0");
        }
    }
    // ---- gather the device-resident stripes: equal (padded) shares, one ncclAllGather per communicator, grouped
    if (m->use_rccl) {
        const RcclApi& rccl = rccl_api();
        NCCLCHK(rccl.GroupStart());
        for (int g = 0; g < G; ++g) {
            ncclResult_t r = rccl.AllGather(m->d_accum[g], m->d_gather[g], (size_t)m->share, ncclFloat, m->comms[g], m->streams[g]);
            if (r != ncclSuccess) { (void)rccl.GroupEnd(); return fail_nccl(r, "ncclAllGather"); }
        }
        NCCLCHK(rccl.GroupEnd());
    } else if (m->loopback && G > 1) {
        // loopback stand-in for the collective: rank g's share -> slot g of the first rank's receive buffer, on rank g's own stream
        // (ordered behind its render, as the collective would be), and the first rank's stream waits for all of them
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipSetDevice(m->devices[g]));
            HIPCHK(hipMemcpyAsync(m->d_gather[0] + (size_t)g * m->share, m->d_accum[g], (size_t)m->share * sizeof(float), hipMemcpyDeviceToDevice, m->streams[g]));
            if (g == 0) continue;
            hipEvent_t ev;
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e1 = hipEventRecord(ev, m->streams[g]);
            hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(m->streams[0], ev, 0) : e1;
            (void)hipEventDestroy(ev);
            if (e2 != hipSuccess) return fail_hip(e2, "loopback gather: event");
        }
    }
    // ---- first device: film order, preview mean, tonemap, copies
    HIPCHK(hipSetDevice(m->devices[0]));
    hipStream_t s0 = m->streams[0];
    hipLaunchKernelGGL(k_unstripe, dim3(blocks_for(n_film)), dim3(256), 0, s0, m->d_gather[0], m->d_film, H, W3, R, G, m->share);
    HIPCHK(hipGetLastError());
    if (out_sums) HIPCHK(hipMemcpyAsync(out_sums, m->d_film, (size_t)n_film * sizeof(float), hipMemcpyDeviceToHost, s0));
    if (out_u8) {
        const int s_done = sample_first + sample_count;
        const float* src = m->d_film;                                  // the last pass divided (main.cpp:126): the sums are the means
        if (s_done < pr->samples) {
            hipLaunchKernelGGL(k_preview_mean, dim3(blocks_for(n_film)), dim3(256), 0, s0, m->d_film, m->d_mean, n_film, static_cast<float>(s_done > 0 ? s_done : 1));
            src = m->d_mean;
        }
        hipLaunchKernelGGL(k_resolve, dim3(blocks_for((long long)W * H)), dim3(256), 0, s0, src, (long long)W * H, m->d_u8);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out_u8, m->d_u8, (size_t)W * H * 3, hipMemcpyDeviceToHost, s0));
    }
    for (int g = G - 1; g >= 0; --g) { HIPCHK(hipSetDevice(m->devices[g])); HIPCHK(hipStreamSynchronize(m->streams[g])); }
    if (stats) {
        hrt_stats total{};
        for (int g = 0; g < G; ++g) {
            hrt_stats one;
            st = hrt_scene_stats(m->scenes[g], &one);
            if (st != HRT_OK) return st;
            total.rays += one.rays; total.samples += one.samples; total.box_tests += one.box_tests; total.tri_tests += one.tri_tests;
            total.mesh_hits += one.mesh_hits; total.env_lookups += one.env_lookups; total.launches += one.launches;
            total.traversal_box_tests += one.traversal_box_tests; total.traversal_tri_tests += one.traversal_tri_tests;
            total.traversal_launches += one.traversal_launches;
            if (one.kernel_ms > total.kernel_ms) total.kernel_ms = one.kernel_ms;          // the devices run side by side
            if (one.traversal_ms > total.traversal_ms) total.traversal_ms = one.traversal_ms;
        }
        *stats = total;
    }
    return HRT_OK;
    HRT_API_CATCH
}

}  // extern "C"
