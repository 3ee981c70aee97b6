// hrt_pack.h — host-side repacking of the flat scene's triangle arrays into the
// 16-byte aligned records the kernels fetch with dwordx4 loads (DScene in
// hrt_device.h), and of its BVH nodes into the compact culling records.  Used by hrt_scene_create (hrt_hip.hip).
#pragma once
#include <cmath>
#include <vector>

#include "../../include/hrt.h"
#include "hrt_glm.h"

namespace hrt {

//   pos : 3 x float4 per triangle  v0.xyz_ v1.xyz_ v2.xyz_
//   attr: 4 x float4 per triangle  n0.xyz uv0.x | n1.xyz uv0.y | n2.xyz uv1.x | uv1.y uv2.x uv2.y _
//   box : 2 x float4 per triangle  accept-box min.xyz + tri_ref_order code (raw bits) | max.xyz _
inline void pack_triangles(const hrt_flat_scene* f, std::vector<float>& pos, std::vector<float>& attr, std::vector<float>& box) {
    const size_t nt = (size_t)f->n_tris;
    pos.assign(nt * 12, 0.0f); attr.assign(nt * 16, 0.0f); box.assign(nt * 8, 0.0f);
    for (size_t i = 0; i < nt; ++i) {
        const float* p = f->tri_pos + 9 * i; const float* n = f->tri_nrm + 9 * i; const float* uv = f->tri_uv + 6 * i;
        for (int k = 0; k < 3; ++k) { pos[12 * i + 4 * k] = p[3 * k]; pos[12 * i + 4 * k + 1] = p[3 * k + 1]; pos[12 * i + 4 * k + 2] = p[3 * k + 2]; }
        float* a = &attr[16 * i];
        a[0] = n[0]; a[1] = n[1]; a[2] = n[2]; a[3] = uv[0];
        a[4] = n[3]; a[5] = n[4]; a[6] = n[5]; a[7] = uv[1];
        a[8] = n[6]; a[9] = n[7]; a[10] = n[8]; a[11] = uv[2];
        a[12] = uv[3]; a[13] = uv[4]; a[14] = uv[5];
        float* b = &box[8 * i];
        if (f->tri_box) {
            const float* s = f->tri_box + 6 * i;
            b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[4] = s[3]; b[5] = s[4]; b[6] = s[5];
        } else {  // triangle.cpp:133-151
            for (int c = 0; c < 3; ++c) {
                float mn = gmin(gmin(p[c], p[3 + c]), p[6 + c]);
                float mx = gmax(gmax(p[c], p[3 + c]), p[6 + c]);
                b[c] = mn - 0.0001f; b[4 + c] = mx + 0.0001f;
            }
        }
        const uint32_t ord = f->tri_ref_order ? f->tri_ref_order[i] : ((uint32_t)i << 1);
        b[3] = u2f(ord);
    }
}

// Culling nodes: the 64-byte fp32 hrt_bvh_node of the ABI becomes a 32-byte record (two dwordx4 fetches per
// step instead of four -- the traversal kernel is bound by the L1 tag pipeline, one divergent lane-access per
// clock) with the child boxes on a 16-bit grid over the mesh's root box:
//   qnode: 8 words  | c0 x lo|hi<<16 | c0 y | c0 z | child0 | c1 x | c1 y | c1 z | child1 |
//   grid : 2 x float4 per mesh  origin.xyz _ | step.xyz _          box = origin + q * step
// Boxes are only ever used to CULL (acceptance uses the reference's exact leaf boxes, tri_box), so any superset
// is valid: lo is rounded down and hi up until the decoded value, computed with the kernel's own fmaf, encloses
// the fp32 box -- and then one cell further.  The kernel evaluates t = q * (step / d) - (o - origin) / d, whose two terms
// are as large as the box's distance from the GRID origin; near a far corner of a mesh with a huge dynamic range they
// cancel and t is only good to ~1.2e-7 x the mesh extent = 0.008 cells (a chain of triangles growing from 1 to 1e10 lost
// 13 % of its hits that way: tests/test_gpu_scenes.py::test_deep_bvh...).  A whole cell of slack (1.5e-5 x extent, the
// size of the reference's own 1e-4 padding for the teapot) covers that with room for ray origins ~50 extents away.
// Empty children (min > max) keep their harmless never-smaller encoding lo=65535, hi=0.
inline void pack_nodes(const hrt_flat_scene* f, std::vector<uint32_t>& qnodes, std::vector<float>& grids) {
    qnodes.assign((size_t)f->n_nodes * 8, 0u);
    grids.assign((size_t)f->n_meshes * 8, 0.0f);
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const hrt_mesh& me = f->meshes[m];
        float* g = &grids[8 * (size_t)m];
        for (int a = 0; a < 3; ++a) { g[a] = 0.0f; g[4 + a] = 1.0f; }
        if (me.node_count == 0) continue;
        const hrt_bvh_node* nodes = f->nodes + me.node_first;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = 0; i < me.node_count; ++i) {   // (children lie inside their parents up to the rounding guard: take all)
            const hrt_bvh_node& n = nodes[i];
            const float b[2][6] = {{n.c0_min_x, n.c0_min_y, n.c0_min_z, n.c0_max_x, n.c0_max_y, n.c0_max_z},
                                   {n.c1_min_x, n.c1_min_y, n.c1_min_z, n.c1_max_x, n.c1_max_y, n.c1_max_z}};
            for (int c = 0; c < 2; ++c) {
                if (b[c][0] > b[c][3]) continue;
                for (int a = 0; a < 3; ++a) { lo[a] = gmin(lo[a], b[c][a]); hi[a] = gmax(hi[a], b[c][3 + a]); }
            }
        }
        for (int a = 0; a < 3; ++a) {
            if (lo[a] > hi[a]) { lo[a] = 0.0f; hi[a] = 0.0f; }
            float step = hi[a] / 65000.0f - lo[a] / 65000.0f;   // (not (hi - lo) / 65000: the extent itself may exceed FLT_MAX)
            // 65000, not 65535 -- head-room: hi must decode to <= 65535 whatever the rounding
            if (!(step > 1e-30f)) step = 1e-30f;
            g[a] = lo[a]; g[4 + a] = step;
        }
        auto q_lo = [&](float v, int a) {
            long q = (long)std::floor(((double)v - (double)g[a]) / (double)g[4 + a]);
            q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
            while (q > 0 && fmaf((float)q, g[4 + a], g[a]) > v) --q;
            if (q > 0) --q;                                   // one more cell: see below
            return (uint32_t)q;
        };
        auto q_hi = [&](float v, int a) {
            long q = (long)std::ceil(((double)v - (double)g[a]) / (double)g[4 + a]);
            q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
            while (q < 65535 && fmaf((float)q, g[4 + a], g[a]) < v) ++q;
            if (q < 65535) ++q;
            return (uint32_t)q;
        };
        for (uint32_t i = 0; i < me.node_count; ++i) {
            const hrt_bvh_node& n = nodes[i];
            uint32_t* w = &qnodes[8 * ((size_t)me.node_first + i)];
            const float b[2][6] = {{n.c0_min_x, n.c0_min_y, n.c0_min_z, n.c0_max_x, n.c0_max_y, n.c0_max_z},
                                   {n.c1_min_x, n.c1_min_y, n.c1_min_z, n.c1_max_x, n.c1_max_y, n.c1_max_z}};
            const int32_t ch[2] = {n.child0, n.child1};
            for (int c = 0; c < 2; ++c) {
                for (int a = 0; a < 3; ++a)
                    w[4 * c + a] = b[c][0] > b[c][3] ? 65535u : (q_lo(b[c][a], a) | (q_hi(b[c][3 + a], a) << 16));
                w[4 * c + 3] = (uint32_t)ch[c];
            }
        }
    }
}

}  // namespace hrt
