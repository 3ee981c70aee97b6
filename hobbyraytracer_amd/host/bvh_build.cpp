// bvh_build.cpp — BVHNode: the reference's pointer-tree BVH (bvh.cpp) rebuilt
// as a flattened array of 64-byte two-child nodes for the HIP traversal.
//
// Topology is free: the closest hit does not depend on it, so
// the tree is built with binned SAH instead of the reference's random-axis
// median split (bvh.cpp:10,39-43).  What DOES depend on the reference's tree is
// which t < t_min self-hits survive (Q-2): BVHNode::hit (bvh.cpp:71) rejects a
// subtree whose box fails AABB::hit(r, t_min, t_max), and the smallest such box
// around a triangle is the one of its lowest BVHNode (itself for a 1-object
// node, the union with its sibling for a 2-object node; bvh.cpp:20-36,52-60).
// referenceLeafBoxes() restates that build to obtain those boxes; the kernel
// applies them to accepted candidates (hrt_device.h accept_box).  The SAH
// tree's own boxes stay tight (union of the padded triangle boxes): the kernel
// culls with the interval [0, closest] when Q-2 is on, so every candidate with
// t > 0 is reached whatever the reference's boxes look like, and accept_box
// then decides exactly as the reference would.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>

#include "../csrc/hrt_rng.h"
#include "classes.h"

namespace hrthost {

namespace {

struct Box3 {
    float mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); } }
    void grow(const Box3& b) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    void growPoint(const float* p) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    float halfArea() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

// ITriangle::boundingBox (triangle.cpp:133-151)
Box3 paddedTriBox(const float* p) {
    Box3 b;
    for (int a = 0; a < 3; ++a) {
        float mn = hrt::gmin(hrt::gmin(p[a], p[3 + a]), p[6 + a]);
        float mx = hrt::gmax(hrt::gmax(p[a], p[3 + a]), p[6 + a]);
        b.mn[a] = mn - 0.0001f;
        b.mx[a] = mx + 0.0001f;
    }
    return b;
}

// bvh.cpp:6-61 restated for topology only.  `order` holds triangle indices.
void refBuild(std::vector<uint32_t>& order, size_t start, size_t end, const std::vector<Box3>& boxes, uint32_t& serial,
              std::vector<float>& out, std::vector<uint32_t>& leafOrder) {
    hrt::u32x4 u = hrt::philox4x32_10(serial++, 0, 0, hrt::RNG_BUILD, 0, 0);  // bvh.cpp:10
    const int a = (int)(u.x % 3u);
    auto comparator = [&](uint32_t x, uint32_t y) { return boxes[x].mn[a] < boxes[y].mn[a]; };  // bvh.cpp:80-90
    const size_t n = end - start;
    auto store = [&](uint32_t tri, const Box3& b) {
        for (int k = 0; k < 3; ++k) { out[6 * tri + k] = b.mn[k]; out[6 * tri + 3 + k] = b.mx[k]; }
    };
    if (n == 1) {
        // left = right = obj ; box = surroundingBox(b, b) = b
        store(order[start], boxes[order[start]]);
        leafOrder[order[start]] = (uint32_t)(start << 1);
    } else if (n == 2) {
        Box3 b;
        for (int k = 0; k < 3; ++k) {  // AABB::surroundingBox (aabb.h:41-56, glm::min / glm::max)
            b.mn[k] = hrt::gmin(boxes[order[start]].mn[k], boxes[order[start + 1]].mn[k]);
            b.mx[k] = hrt::gmax(boxes[order[start]].mx[k], boxes[order[start + 1]].mx[k]);
        }
        // bvh.cpp:26-36: left = the lesser on the chosen axis, so the walk order inside the pair may swap
        const bool keep = comparator(order[start], order[start + 1]);
        store(order[start], b);
        store(order[start + 1], b);
        // code = (leaf-level node id << 1) | side, side 0 = `left` (tested first), 1 = `right`
        leafOrder[order[start]] = (uint32_t)((start << 1) | (keep ? 0u : 1u));
        leafOrder[order[start + 1]] = (uint32_t)((start << 1) | (keep ? 1u : 0u));
    } else if (n != 0) {
        std::sort(order.begin() + start, order.begin() + end, comparator);  // bvh.cpp:39
        const size_t mid = start + n / 2;
        refBuild(order, start, mid, boxes, serial, out, leafOrder);
        refBuild(order, mid, end, boxes, serial, out, leafOrder);
    }
}

struct SahBuilder {
    struct Ref { Box3 b; float c[3]; uint32_t tri; };
    std::vector<Ref> refs;
    std::vector<hrt_bvh_node> nodes;
    int maxLeaf = 2;     // measured on MI355X (tests/tools/sweep_bvh.py): 2 beats 1, 3, 4, 8 by 3-12 % of traversal time on teapot and bust
    int sahLevels = 31;  // depth budget: SAH only while depth + ceil(log2 n) + 1 < sahLevels
    int maxDepth = 0;
    float triCost = 1.3f, boxCost = 1.0f;

    static void setChild(hrt_bvh_node& n, int c, const Box3& b, int32_t ref) {
        if (c == 0) { n.c0_min_x = b.mn[0]; n.c0_max_x = b.mx[0]; n.c0_min_y = b.mn[1]; n.c0_max_y = b.mx[1]; n.c0_min_z = b.mn[2]; n.c0_max_z = b.mx[2]; n.child0 = ref; }
        else { n.c1_min_x = b.mn[0]; n.c1_max_x = b.mx[0]; n.c1_min_y = b.mn[1]; n.c1_max_y = b.mx[1]; n.c1_min_z = b.mn[2]; n.c1_max_z = b.mx[2]; n.child1 = ref; }
    }
    static int32_t leafRef(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1)); }

    int32_t build(uint32_t lo, uint32_t hi, int depth, Box3& outBox) {
        const uint32_t n = hi - lo;
        outBox.reset();
        Box3 cb; cb.reset();
        for (uint32_t i = lo; i < hi; ++i) { outBox.grow(refs[i].b); cb.growPoint(refs[i].c); }
        if (depth > maxDepth) maxDepth = depth;
        if (n == 1) return leafRef(lo, 1);

        // choose the split
        uint32_t mid = lo;
        bool haveSplit = false;
        int axis = 0;
        { float ext = -1; for (int a = 0; a < 3; ++a) { float e = cb.mx[a] - cb.mn[a]; if (e > ext) { ext = e; axis = a; } } }
        const float leafCost = triCost * n;
        // SAH while the remaining balanced depth still fits the kernel's stack
        int lgn = 0; while ((1u << lgn) < n) ++lgn;
        const bool sahAllowed = depth + lgn + 1 < sahLevels;
        if (sahAllowed && n > 2) {
            const int NB = 16;
            float bestCost = std::numeric_limits<float>::infinity(); int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                const float e = cb.mx[a] - cb.mn[a];
                if (!(e > 0)) continue;
                Box3 bb[NB]; uint32_t bc[NB];
                for (int k = 0; k < NB; ++k) { bb[k].reset(); bc[k] = 0; }
                const float scale = NB / e;
                for (uint32_t i = lo; i < hi; ++i) {
                    int k = (int)((refs[i].c[a] - cb.mn[a]) * scale);
                    k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                    bb[k].grow(refs[i].b); bc[k]++;
                }
                float rightArea[NB]; uint32_t rightCount[NB];
                Box3 acc; acc.reset(); uint32_t cnt = 0;
                for (int k = NB - 1; k > 0; --k) { acc.grow(bb[k]); cnt += bc[k]; rightArea[k] = acc.halfArea(); rightCount[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = 0; k < NB - 1; ++k) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (cnt == 0 || rightCount[k + 1] == 0) continue;
                    const float cost = acc.halfArea() * cnt + rightArea[k + 1] * rightCount[k + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
                }
            }
            if (bestAxis >= 0) {
                const float parentArea = outBox.halfArea();
                const float splitCost = 2 * boxCost + triCost * bestCost / (parentArea > 0 ? parentArea : 1.0f);
                if (n <= (uint32_t)maxLeaf && leafCost <= splitCost) return leafRef(lo, n);
                const float e = cb.mx[bestAxis] - cb.mn[bestAxis];
                const float scale = 16 / e;
                auto it = std::partition(refs.begin() + lo, refs.begin() + hi, [&](const Ref& r) {
                    int k = (int)((r.c[bestAxis] - cb.mn[bestAxis]) * scale);
                    k = k < 0 ? 0 : (k >= 16 ? 15 : k);
                    return k <= bestBin;
                });
                mid = (uint32_t)(it - refs.begin());
                haveSplit = mid > lo && mid < hi;
            }
        }
        if (!haveSplit) {
            if (n <= (uint32_t)maxLeaf) return leafRef(lo, n);
            // balanced median split on the widest centroid axis (bounds the depth)
            mid = lo + n / 2;
            std::nth_element(refs.begin() + lo, refs.begin() + mid, refs.begin() + hi,
                             [axis](const Ref& x, const Ref& y) { return x.c[axis] < y.c[axis]; });
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(hrt_bvh_node{});
        Box3 b0, b1;
        const int32_t c0 = build(lo, mid, depth + 1, b0);
        const int32_t c1 = build(mid, hi, depth + 1, b1);
        hrt_bvh_node nd{};
        setChild(nd, 0, b0, c0);
        setChild(nd, 1, b1, c1);
        nodes[me] = nd;
        return me;
    }
};

void emptyChild(hrt_bvh_node& n, int c) {
    const float inf = std::numeric_limits<float>::infinity();
    if (c == 0) { n.c0_min_x = n.c0_min_y = n.c0_min_z = inf; n.c0_max_x = n.c0_max_y = n.c0_max_z = -inf; n.child0 = -1; }
    else { n.c1_min_x = n.c1_min_y = n.c1_min_z = inf; n.c1_max_x = n.c1_max_y = n.c1_max_z = -inf; n.child1 = -1; }
}

// Refit: every child box becomes the union of the padded ITriangle boxes
// (triangle.cpp:133-151) of the triangles below it, widened by a rounding guard
// so that the fma-based slab test of the kernel can never be tighter than a
// division-based test of the same box.
Box3 refit(std::vector<hrt_bvh_node>& nodes, int32_t ref, const std::vector<float>& pos, int depth, int& maxDepth) {
    Box3 b; b.reset();
    if (ref < 0) {
        const uint32_t enc = (uint32_t)~ref;
        const uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
        for (uint32_t k = 0; k < count; ++k) {
            b.grow(paddedTriBox(&pos[9 * (first + k)]));
        }
        return b;
    }
    if (depth > maxDepth) maxDepth = depth;
    hrt_bvh_node& n = nodes[ref];
    auto guard = [](Box3 x) {
        for (int a = 0; a < 3; ++a) {
            const float g = 1e-6f + 4e-7f * std::max(std::fabs(x.mn[a]), std::fabs(x.mx[a]));
            x.mn[a] -= g; x.mx[a] += g;
        }
        return x;
    };
    const bool e0 = n.c0_min_x > n.c0_max_x, e1 = n.c1_min_x > n.c1_max_x;
    if (!e0) {
        Box3 c = refit(nodes, n.child0, pos, depth + 1, maxDepth);
        b.grow(c); c = guard(c);
        n.c0_min_x = c.mn[0]; n.c0_max_x = c.mx[0]; n.c0_min_y = c.mn[1]; n.c0_max_y = c.mx[1]; n.c0_min_z = c.mn[2]; n.c0_max_z = c.mx[2];
    }
    if (!e1) {
        Box3 c = refit(nodes, n.child1, pos, depth + 1, maxDepth);
        b.grow(c); c = guard(c);
        n.c1_min_x = c.mn[0]; n.c1_max_x = c.mx[0]; n.c1_min_y = c.mn[1]; n.c1_max_y = c.mx[1]; n.c1_min_z = c.mn[2]; n.c1_max_z = c.mx[2];
    }
    return b;
}

}  // namespace

// Restates bvh.cpp:6-61 over the soup (in its CURRENT order) and returns, per
// triangle, the box of its lowest BVHNode and (leafOrder) a code
// (node << 1) | side: `node` numbers the lowest BVHNodes in the tree's
// left-to-right order, `side` says whether the triangle is that node's `left`
// (0, tested first) or `right` (1) child (bvh.cpp:20-36).
std::vector<float> referenceLeafBoxes(const TriangleSoup& soup, std::vector<uint32_t>& leafOrder) {
    const size_t n = soup.size();
    std::vector<Box3> boxes(n);
    for (size_t i = 0; i < n; ++i) boxes[i] = paddedTriBox(&soup.pos[9 * i]);
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::vector<float> out(6 * n);
    leafOrder.assign(n, 0);
    uint32_t serial = 0;
    if (n) refBuild(order, 0, n, boxes, serial, out, leafOrder);
    return out;
}

namespace {
typedef hrt_status (*DeviceBuildFn)(int, const float*, uint32_t, uint32_t, hrt_bvh_node*, uint32_t*, uint32_t*, int32_t*);
DeviceBuildFn g_deviceBuild = nullptr;
int g_deviceBuildDevice = 0;
}  // namespace
void setDeviceBvhBuilder(void* fn, int device) { g_deviceBuild = (DeviceBuildFn)fn; g_deviceBuildDevice = device; }

BVHNode::BVHNode(TriangleSoup& soup) {
    const size_t n = soup.size();
    if (n >= (1u << 28)) throw FlattenError(HRT_ERR_UNSUPPORTED, "mesh has too many triangles");
    if (n == 0) {  // mesh.cpp:21,38 with a failed import: an empty tree that never hits
        hrt_bvh_node root{};
        emptyChild(root, 0); emptyChild(root, 1);
        nodes.push_back(root);
        depth = 1;
        return;
    }
    SahBuilder sb;
    // tuning knobs (experiments only; the defaults are what the tests and the bench use)
    if (const char* e = std::getenv("HRT_BVH_MAX_LEAF")) sb.maxLeaf = std::min(8, std::max(1, std::atoi(e)));
    if (g_deviceBuild && n > (size_t)sb.maxLeaf) {
        // the tree from the GPU (csrc/hrt_lbvh.hip): topology, leaf order, boxes and depth come back finished
        std::vector<hrt_bvh_node> dn(n - 1);
        std::vector<uint32_t> order(n);
        uint32_t nNodes = 0; int32_t dDepth = 0;
        const hrt_status st = g_deviceBuild(g_deviceBuildDevice, soup.pos.data(), (uint32_t)n, (uint32_t)sb.maxLeaf, dn.data(), &nNodes, order.data(), &dDepth);
        if (st != HRT_OK && st != HRT_ERR_UNSUPPORTED) throw FlattenError(st, "the device BVH builder failed (hrt_last_error() of libhrt_hip.so has the reason)");
        if (st == HRT_ERR_UNSUPPORTED)     // hrt_bvh_build_sah met a large node that needs the median split of the code below
            std::cerr << "note: the GPU builder hands a mesh of " << n << " triangles back to the host SAH builder" << std::endl;
        else if (dDepth <= 31 && nNodes != 0) {
        dn.resize(nNodes);
        TriangleSoup re;
        re.pos.resize(9 * n); re.nrm.resize(9 * n); re.uv.resize(6 * n);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t src = order[i];
            if (src >= n) throw FlattenError(HRT_ERR_INVALID, "the device BVH builder returned a bad leaf order");
            std::memcpy(&re.pos[9 * i], &soup.pos[9 * src], 36);
            std::memcpy(&re.nrm[9 * i], &soup.nrm[9 * src], 36);
            std::memcpy(&re.uv[6 * i], &soup.uv[6 * src], 24);
        }
        soup = std::move(re);
        nodes = std::move(dn);
        leafBoxes = referenceLeafBoxes(soup, refOrder);
        depth = dDepth;
        return;
        }
        // a tree deeper than the traversal's stack (31 levels: clustered geometry inside a huge bound can do that to a Morton
        // tree) is not truncated and not a reason to fail the load: this mesh gets the host's SAH tree, whose depth is bounded
        else std::cerr << "note: the GPU-built BVH of a mesh of " << n << " triangles is " << dDepth << " levels deep (the traversal stack holds 31): "
                     "building this mesh's tree with the host SAH builder instead" << std::endl;
    }
    if (const char* e = std::getenv("HRT_BVH_TRI_COST")) sb.triCost = (float)std::atof(e);
    sb.refs.resize(n);
    for (size_t i = 0; i < n; ++i) {
        SahBuilder::Ref& r = sb.refs[i];
        r.b = paddedTriBox(&soup.pos[9 * i]);
        for (int a = 0; a < 3; ++a) r.c[a] = 0.5f * (r.b.mn[a] + r.b.mx[a]);
        r.tri = (uint32_t)i;
    }
    // total depth stays within the kernel's LDS stack (HRT_STACK_DEPTH = 32): see SahBuilder::sahLevels
    Box3 rootBox;
    const int32_t rootRef = sb.build(0, (uint32_t)n, 1, rootBox);
    if (rootRef < 0) {  // the whole mesh fits one leaf: wrap it in a root node
        hrt_bvh_node root{};
        SahBuilder::setChild(root, 0, rootBox, rootRef);
        emptyChild(root, 1);
        sb.nodes.push_back(root);
    }
    // reorder the soup into leaf order
    TriangleSoup re;
    re.pos.resize(9 * n); re.nrm.resize(9 * n); re.uv.resize(6 * n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t src = sb.refs[i].tri;
        std::memcpy(&re.pos[9 * i], &soup.pos[9 * src], 36);
        std::memcpy(&re.nrm[9 * i], &soup.nrm[9 * src], 36);
        std::memcpy(&re.uv[6 * i], &soup.uv[6 * src], 24);
    }
    soup = std::move(re);
    nodes = std::move(sb.nodes);
    leafBoxes = referenceLeafBoxes(soup, refOrder);
    int md = 1;
    refit(nodes, 0, soup.pos, 1, md);
    depth = md;
}

}  // namespace hrthost
