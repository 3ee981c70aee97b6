// flat_on_cpu.cpp — TEST TOOL ONLY (built by tests/tools/Makefile into tests/tools/libflatcpu.so).
//
// Compiles the product's device header (hobbyraytracer_amd/csrc/hrt_device.h) for the HOST so that the
// CPU-only test run (-m "not gpu") can check what does not need a GPU to be wrong: the flattened BVH
// produced by the host builder, the acceptance rules of bvh_traverse (leaf boxes, self-hit order) and the
// per-segment code, all against the oracle on millions of seeded rays.  It is also the quick way to
// localise a rare GPU/oracle mismatch.  It is NOT part of the product: nothing under hobbyraytracer_amd/
// or include/ references it, it is never installed into lib/, and the product has no CPU render path.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __device__
struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }

#include "../../hobbyraytracer_amd/csrc/hrt_device.h"
#include "../../hobbyraytracer_amd/csrc/hrt_pack.h"

using namespace hrt;

namespace {
struct CpuScene {
    std::vector<float> pos, attr, box, grids;
    std::vector<uint32_t> qnodes;
    RefTree ref;
    DScene ds;
};
CpuScene* make(const hrt_flat_scene* f) {
    CpuScene* s = new CpuScene;
    if (!pack_ref_tree(f, s->ref)) { delete s; return nullptr; }
    pack_triangles(f, s->ref, s->pos, s->attr, s->box);
    s->ds.rnodes = (const uint4*)s->ref.nodes.data(); s->ds.rtris = (const float4*)s->ref.tris.data(); s->ds.rmesh = (const uint4*)s->ref.mesh_nodes.data();
    {
        float a = HRT_Q4_ROUTE_A_DEFAULT;
        if (const char* e = getenv("HRT_Q4_ROUTE_A")) a = (float)atof(e);
        s->ds.q4_route_a2 = a * a;
        s->ds.ref_fold_all = getenv("HRT_REF_FOLD_ALL") ? 1 : 0;
        s->ds.stale_ff = scene_has_stale_front_face(f);
    }
    s->ds.prims = f->prims; s->ds.mats = f->materials; s->ds.texs = f->textures; s->ds.meshes = f->meshes;
    pack_nodes(f, s->qnodes, s->grids);
    s->ds.qnodes = (const uint4*)s->qnodes.data(); s->ds.grids = (const float4*)s->grids.data();
    s->ds.tri_pos = (const float4*)s->pos.data(); s->ds.tri_attr = (const float4*)s->attr.data(); s->ds.tri_box = (const float4*)s->box.data();
    s->ds.texels_u8 = f->texels_u8; s->ds.texels_f32 = f->texels_f32;
    s->ds.n_prims = (int32_t)f->n_prims; s->ds.background_tex = f->background_tex;
    s->ds.lprims = f->prims; s->ds.lmats = f->materials; s->ds.ltexs = f->textures; s->ds.lmeshes = f->meshes;
    s->ds.n_mats = (int32_t)f->n_materials; s->ds.n_texs = (int32_t)f->n_textures; s->ds.n_meshes = (int32_t)f->n_meshes;
    return s;
}
}  // namespace

extern "C" {

void* flatcpu_create(const hrt_flat_scene* f) { return make(f); }
void flatcpu_destroy(void* h) { delete (CpuScene*)h; }

// the packed culling records the product uploads (hrt_pack.h pack_nodes): 8 words per node, 8 floats per mesh
const uint32_t* flatcpu_qnodes(void* h, uint64_t* n_words) { CpuScene* s = (CpuScene*)h; *n_words = s->qnodes.size(); return s->qnodes.data(); }
const float* flatcpu_grids(void* h, uint64_t* n_floats) { CpuScene* s = (CpuScene*)h; *n_floats = s->grids.size(); return s->grids.data(); }

void flatcpu_closest_hit(void* h, const hrt_params* pr, int64_t n, const float* ro, const float* rd, float t_min, float t_max,
                         uint32_t pixel0, hrt_hit* out) {
    DScene sc = ((CpuScene*)h)->ds;
    sc.stale_ff = 1;      // as k_hits: the record always carries the inherited frontFace
    std::vector<int> stack(HRT_STACK_DEPTH * HRT_BLOCK);
    for (int64_t i = 0; i < n; ++i) {
        const vec3 o(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), d(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        rng_ctx ctx; ctx.seed_lo = pr->seed_lo; ctx.seed_hi = pr->seed_hi; ctx.pixel = pixel0 + (uint32_t)i; ctx.sample = 0; ctx.bounce = 0;
        DCounters cnt; cnt.box_tests = 0; cnt.tri_tests = 0;
        const WorldHit wh = world_hit<false>(sc, o, d, t_min, t_max, pr->quirks, ctx, stack.data(), cnt);
        hrt_hit hh; std::memset(&hh, 0, sizeof(hh));
        hh.prim = wh.prim; hh.tri = -1;
        if (wh.prim >= 0) {
            DRec rec;
            hit_record(sc, wh, o, d, pr->quirks, t_min, rec);
            hh.t = rec.t; hh.tri = sc.prims[wh.prim].kind == HRT_PRIM_MESH ? (wh.sub & ~(HRT_SUB_WRAPPERLESS | HRT_SUB_STALE_BACK)) : -1; hh.front_face = rec.frontFace ? 1 : 0;
            hh.p[0] = rec.p.x; hh.p[1] = rec.p.y; hh.p[2] = rec.p.z;
            hh.normal[0] = rec.normal.x; hh.normal[1] = rec.normal.y; hh.normal[2] = rec.normal.z;
            hh.u = rec.u; hh.v = rec.v;
        }
        out[i] = hh;
    }
}

// The per-lane logic of k_pathtrace, one pixel after the other.
void flatcpu_render_tile(void* h, const hrt_camera* cam, const hrt_params* pr, hrt_rect tile, float* out, hrt_stats* stats) {
    const DScene& sc = ((CpuScene*)h)->ds;
    std::vector<int> stack(HRT_STACK_DEPTH * HRT_BLOCK);
    PathCounters pc; std::memset(&pc, 0, sizeof(pc));
    for (int ry = 0; ry < tile.h; ++ry)
        for (int rx = 0; rx < tile.w; ++rx) {
            const int px = tile.x0 + rx, py = tile.y0 + ry;
            vec3 sum(0.0f);
            for (int s = 0; s < pr->samples; ++s) {
                rng_ctx ctx; ctx.seed_lo = pr->seed_lo; ctx.seed_hi = pr->seed_hi; ctx.pixel = (uint32_t)(py * pr->width + px); ctx.sample = (uint32_t)s; ctx.bounce = 0;
                PathState ps;
                path_begin(*cam, *pr, px, py, ctx, ps);
                pc.samples++;
                while (!path_segment<true>(sc, *pr, ctx, ps, stack.data(), pc)) {}
                sum += ps.result;
            }
            const vec3 mean = sum / static_cast<float>(pr->samples);
            float* o = out + 3 * ((size_t)ry * tile.w + rx);
            o[0] = mean.x; o[1] = mean.y; o[2] = mean.z;
        }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = pc.rays; stats->samples = pc.samples; stats->box_tests = pc.bvh.box_tests; stats->tri_tests = pc.bvh.tri_tests;
        stats->mesh_hits = pc.mesh_hits; stats->env_lookups = pc.env_lookups;
    }
}

}  // extern "C"
