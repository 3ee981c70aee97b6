// hrt_pack.h — host-side repacking of the flat scene's triangle arrays into the
// 16-byte aligned records the kernels fetch with dwordx4 loads (DScene in
// hrt_device.h), and of its BVH nodes into the compact culling records.  Used by hrt_scene_create (hrt_hip.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/hrt.h"
#include "hrt_glm.h"

namespace hrt {

// ITriangle::boundingBox (triangle.cpp:133-151): min / max of the vertices, padded by 0.0001.
inline void padded_tri_box(const float* p /* 9 floats */, float* mn, float* mx) {
    for (int c = 0; c < 3; ++c) {
        mn[c] = gmin(gmin(p[c], p[3 + c]), p[6 + c]) - 0.0001f;
        mx[c] = gmax(gmax(p[c], p[3 + c]), p[6 + c]) + 0.0001f;
    }
}

// The REFERENCE's own tree of every mesh (bvh.cpp:6-61), for the rays that must be walked through it node by node
// (hrt_device.h ref_walk: quirk Q-4 with a vanishing direction component on the shear axis, where ITriangle::hit's t no
// longer agrees with the boxes and the winner depends on that tree's visiting order).
// The tree is not shipped over the ABI, it is implied by what is: the reference sorts and splits at start + n / 2
// (bvh.cpp:39-43) until one or two objects are left (bvh.cpp:20-36), so a node is a contiguous range of the depth-first
// triangle order -- hrt_flat_scene::tri_ref_order holds each triangle's (range start << 1) | side -- and its box is the
// union of the padded triangle boxes of its range (bvh.cpp:52-60: surroundingBox is min / max, exact and associative).
//   nodes: 8 words per BVHNode, in PREORDER (= the order BVHNode::hit visits them, bvh.cpp:69-78: box, left, right), threaded:
//          | min.xyz  skip | max.xyz  leaf |   skip = the node that follows this node's subtree (mesh-local; == node count
//          at the end); the node after an inner node whose box passed is simply the next one.
//          leaf = 0x80000000 | index of the `right` child for an inner node (`left` is the next node), else
//          (first << 1) | (count - 1): `count` (1 or 2) triangles at positions first, first + 1 of `tris` (mesh-local;
//          fewer than 2^28 triangles per mesh), `left` first (bvh.cpp:74-75).  (A one-object node has left == right,
//          bvh.cpp:20-23, and tests its triangle twice; the second test cannot change the record and is left out.)
//   tris : 3 x float4 per triangle in the reference's TEST order: v0.xyz | mesh-local triangle index (bits), v1.xyz_, v2.xyz_
//   mesh_nodes: 4 words per mesh: first node, node count, first triangle of the mesh in `tris`, 0
// tri_ref_order == NULL: the triangles are taken in the given order (a reference tree whose sorts changed nothing);
// `codes` / `leaf_box` then hold what tri_ref_order / tri_box would have said (pack_triangles uses them).
struct RefTree {
    std::vector<uint32_t> nodes;
    std::vector<float> tris;
    std::vector<uint32_t> mesh_nodes;
    std::vector<uint32_t> codes;      // per triangle (global index)
    std::vector<float> leaf_box;      // 6 floats per triangle: the box of its lowest BVHNode
};
namespace detail {
struct RefBuilder {
    const float* pos;                 // mesh's tri_pos (9 floats per triangle)
    const std::vector<uint32_t>* ord; // depth-first order: mesh-local triangle indices
    std::vector<uint32_t>* nodes;
    uint32_t first_node;              // of this mesh, in *nodes (8 words each)
    // builds the subtree over ord[start, start + n); returns its box in bmn / bmx
    void build(uint32_t start, uint32_t n, float* bmn, float* bmx) {
        const size_t me = nodes->size() / 8;
        nodes->resize(nodes->size() + 8, 0u);
        uint32_t leaf;
        if (n <= 2) {
            padded_tri_box(pos + 9 * (size_t)(*ord)[start], bmn, bmx);
            if (n == 2) {   // AABB::surroundingBox (aabb.h:41-56)
                float mn2[3], mx2[3];
                padded_tri_box(pos + 9 * (size_t)(*ord)[start + 1], mn2, mx2);
                for (int c = 0; c < 3; ++c) { bmn[c] = gmin(bmn[c], mn2[c]); bmx[c] = gmax(bmx[c], mx2[c]); }
            }
            leaf = (start << 1) | (n - 1);
        } else {
            float lmn[3], lmx[3], rmn[3], rmx[3];
            build(start, n / 2, lmn, lmx);
            leaf = 0x80000000u | (uint32_t)(nodes->size() / 8 - first_node);   // where `right` is about to be put
            build(start + n / 2, n - n / 2, rmn, rmx);
            for (int c = 0; c < 3; ++c) { bmn[c] = gmin(lmn[c], rmn[c]); bmx[c] = gmax(lmx[c], rmx[c]); }
        }
        uint32_t* w = nodes->data() + 8 * me;
        for (int c = 0; c < 3; ++c) { w[c] = f2u(bmn[c]); w[4 + c] = f2u(bmx[c]); }
        w[3] = (uint32_t)(nodes->size() / 8 - first_node);
        w[7] = leaf;
    }
};
}  // namespace detail
// false (+ *err): tri_ref_order is not the depth-first code of such a tree.
inline bool pack_ref_tree(const hrt_flat_scene* f, RefTree& out, std::string* err = nullptr) {
    const size_t nt = (size_t)f->n_tris;
    out.nodes.clear(); out.tris.clear(); out.mesh_nodes.assign((size_t)f->n_meshes * 4, 0u);
    out.codes.assign(nt, 0u); out.leaf_box.assign(nt * 6, 0.0f);
    for (uint32_t m = 0; m < f->n_meshes; ++m) {
        const hrt_mesh& me = f->meshes[m];
        const uint32_t n = me.tri_count;
        out.mesh_nodes[4 * m] = (uint32_t)(out.nodes.size() / 8);
        const size_t tri_base = out.tris.size() / 12;   // (meshes may share triangle ranges: every mesh gets its own copy)
        out.mesh_nodes[4 * m + 2] = (uint32_t)tri_base;
        if (n == 0) continue;
        out.tris.resize(out.tris.size() + 12 * (size_t)n, 0.0f);
        std::vector<uint32_t> ord(n);
        for (uint32_t i = 0; i < n; ++i) ord[i] = i;
        if (f->tri_ref_order) {
            const uint32_t* code = f->tri_ref_order + me.tri_first;
            std::sort(ord.begin(), ord.end(), [code](uint32_t a, uint32_t b) { return code[a] < code[b]; });
        }
        detail::RefBuilder rb;
        rb.pos = f->tri_pos + 9 * (size_t)me.tri_first; rb.ord = &ord; rb.nodes = &out.nodes; rb.first_node = out.mesh_nodes[4 * m];
        float mn[3], mx[3];
        rb.build(0, n, mn, mx);
        const uint32_t count = (uint32_t)(out.nodes.size() / 8) - rb.first_node;
        out.mesh_nodes[4 * m + 1] = count;
        // the leaves: codes, leaf-level boxes, triangles in test order
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t* w = out.nodes.data() + 8 * ((size_t)rb.first_node + i);
            if (w[7] & 0x80000000u) continue;
            const uint32_t first = w[7] >> 1, cnt = (w[7] & 1u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t tri = ord[first + k];
                const uint32_t code = (first << 1) | k;
                if (f->tri_ref_order && f->tri_ref_order[me.tri_first + tri] != code) {
                    if (err) *err = "tri_ref_order is not the depth-first code of a median-split tree (mesh " + std::to_string(m) + ", triangle " + std::to_string(tri) + ")";
                    return false;
                }
                const size_t g = (size_t)me.tri_first + tri;
                out.codes[g] = code;
                for (int c = 0; c < 3; ++c) { out.leaf_box[6 * g + c] = u2f(w[c]); out.leaf_box[6 * g + 3 + c] = u2f(w[4 + c]); }
                float* t = &out.tris[12 * (tri_base + first + k)];
                const float* p = rb.pos + 9 * (size_t)tri;
                for (int v = 0; v < 3; ++v) { t[4 * v] = p[3 * v]; t[4 * v + 1] = p[3 * v + 1]; t[4 * v + 2] = p[3 * v + 2]; }
                t[3] = u2f(tri);
            }
        }
    }
    return true;
}

//   pos : 3 x float4 per triangle  v0.xyz_ v1.xyz_ v2.xyz_
//   attr: 4 x float4 per triangle  n0.xyz uv0.x | n1.xyz uv0.y | n2.xyz uv1.x | uv1.y uv2.x uv2.y _
//   box : 2 x float4 per triangle  accept-box min.xyz + tri_ref_order code (raw bits) | max.xyz _
// `ref` (pack_ref_tree's result) supplies the codes / leaf-level boxes the scene did not bring (tri_ref_order / tri_box == NULL).
inline void pack_triangles(const hrt_flat_scene* f, const RefTree& ref, std::vector<float>& pos, std::vector<float>& attr, std::vector<float>& box) {
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
        const float* s = f->tri_box ? f->tri_box + 6 * i : &ref.leaf_box[6 * i];
        b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[4] = s[3]; b[5] = s[4]; b[6] = s[5];
        b[3] = u2f(ref.codes[i]);
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
// Node order inside a mesh.  The builders emit depth-first preorder (child0 of node i is node i + 1), so a walk that keeps
// turning towards child0 stays in one 128-byte line (4 records) and every turn towards child1 leaves it.  layout == 1 packs
// TREELETS instead: a line is filled with a subtree chosen greedily by box area (the nodes a random ray is most likely to need
// next), its leftover frontier nodes start the following lines, largest first.  The root stays node 0; only indices change.
// perm[old] = new.
inline void treelet_order(const hrt_bvh_node* nodes, uint32_t n, std::vector<uint32_t>& perm) {
    perm.assign(n, 0xffffffffu);
    if (n == 0) return;
    auto area = [&](uint32_t i) {
        const hrt_bvh_node& d = nodes[i];
        float lo[3] = {gmin(d.c0_min_x, d.c1_min_x), gmin(d.c0_min_y, d.c1_min_y), gmin(d.c0_min_z, d.c1_min_z)};
        float hi[3] = {gmax(d.c0_max_x, d.c1_max_x), gmax(d.c0_max_y, d.c1_max_y), gmax(d.c0_max_z, d.c1_max_z)};
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return (x > 0 && y > 0 && z > 0) ? x * y + y * z + z * x : 0.0f;
    };
    uint32_t next = 0;
    std::vector<uint32_t> roots(1, 0u);          // a stack of treelet roots; the most probable is taken first
    std::vector<uint32_t> frontier;
    while (!roots.empty()) {
        const uint32_t root = roots.back(); roots.pop_back();
        if (perm[root] != 0xffffffffu) continue;
        frontier.assign(1, root);
        int placed = 0;
        while (placed < 4 && !frontier.empty()) {
            size_t best = 0;
            for (size_t k = 1; k < frontier.size(); ++k) if (area(frontier[k]) > area(frontier[best])) best = k;
            const uint32_t i = frontier[best];
            frontier.erase(frontier.begin() + (long)best);
            perm[i] = next++; ++placed;
            if (nodes[i].child0 >= 0 && (uint32_t)nodes[i].child0 < n) frontier.push_back((uint32_t)nodes[i].child0);
            if (nodes[i].child1 >= 0 && (uint32_t)nodes[i].child1 < n) frontier.push_back((uint32_t)nodes[i].child1);
        }
        // what is left starts new lines: smallest pushed first, so that the largest is on top of the stack
        std::sort(frontier.begin(), frontier.end(), [&](uint32_t a, uint32_t b) { return area(a) < area(b); });
        for (uint32_t i : frontier) roots.push_back(i);
    }
    for (uint32_t i = 0; i < n; ++i) if (perm[i] == 0xffffffffu) perm[i] = next++;      // (unreachable nodes: validation refuses them anyway)
}

inline void pack_nodes(const hrt_flat_scene* f, std::vector<uint32_t>& qnodes, std::vector<float>& grids, int layout = 0) {
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
        std::vector<uint32_t> perm;
        if (layout == 1) treelet_order(nodes, me.node_count, perm);
        for (uint32_t i = 0; i < me.node_count; ++i) {
            const hrt_bvh_node& n = nodes[i];
            uint32_t* w = &qnodes[8 * ((size_t)me.node_first + (perm.empty() ? i : perm[i]))];
            const float b[2][6] = {{n.c0_min_x, n.c0_min_y, n.c0_min_z, n.c0_max_x, n.c0_max_y, n.c0_max_z},
                                   {n.c1_min_x, n.c1_min_y, n.c1_min_z, n.c1_max_x, n.c1_max_y, n.c1_max_z}};
            int32_t ch[2] = {n.child0, n.child1};
            for (int c = 0; c < 2; ++c) {
                if (!perm.empty() && ch[c] >= 0 && (uint32_t)ch[c] < me.node_count) ch[c] = (int32_t)perm[(uint32_t)ch[c]];
                for (int a = 0; a < 3; ++a)
                    w[4 * c + a] = b[c][0] > b[c][3] ? 65535u : (q_lo(b[c][a], a) | (q_hi(b[c][3 + a], a) << 16));
                w[4 * c + 3] = (uint32_t)ch[c];
            }
        }
    }
}

// DScene::stale_ff: is there a mesh that stands in the world list without a wrapper and has a Dielectric material?
inline int scene_has_stale_front_face(const hrt_flat_scene* f) {
    for (uint32_t i = 0; i < f->n_prims; ++i) {
        const hrt_prim& p = f->prims[i];
        if (p.kind == HRT_PRIM_MESH && p.n_xforms == 0 && p.material >= 0 && (uint32_t)p.material < f->n_materials && f->materials[p.material].kind == HRT_MAT_DIELECTRIC) return 1;
    }
    return 0;
}

}  // namespace hrt
