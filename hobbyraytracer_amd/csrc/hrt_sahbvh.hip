// hrt_sahbvh.hip -- hrt_bvh_build_sah: the binned-SAH culling tree of host/bvh_build.cpp (SahBuilder::build), built on the GPU.
//
// What it replaces: the BVHNode constructor (bvh.cpp:6-61) as far as the TOPOLOGY of the flattened tree goes; the closest hit
// does not depend on it.  Round 2's GPU builder (hrt_lbvh.hip: a Morton LBVH) costs +16 % box tests per
// ray, and clustering the Morton order bottom-up by surface area (PLOC, tried in round 3) is no better on these meshes
// (teapot 20.6 box tests per segment against 20.4 for the LBVH and 17.5 for the SAH tree).  So this file runs the HOST's
// algorithm itself on the device -- same decisions, same arithmetic (16 bins over the centroid bounds on each axis, first
// strictly better cost wins, leaves of <= max_leaf triangles, -ffp-contract=off), hence the same tree up to the order of the
// triangles inside a leaf:
//
//   large nodes (more than HRT_SAH_SMALL triangles), level by level:
//     k_sah_plan     one block: cuts every active node into chunks of HRT_SAH_CHUNK triangles, clears its statistics
//     k_sah_bounds   block per chunk: box of the padded ITriangle boxes (triangle.cpp:133-151) and of their centroids
//     k_sah_bins     block per chunk: 3 x 16 bins (box + count) in LDS, folded into the node's with ordered-integer atomics
//     k_sah_eval     thread per node: the sweep over the bins, leaf / split decision, node allocation, the two children
//     k_sah_count / k_sah_scatter   block per chunk: partition of the node's index range by "bin <= best bin"
//   small nodes: k_sah_small, one WAVE per subtree, the host's recursion with an explicit stack; the subtree's triangles sit in
//     LDS and the 64 lanes share every pass over them (one thread per subtree took 8 ms for the 100 k-triangle bust: a chain
//     of ~4000 dependent loads per thread)
//
// A node whose SAH is not allowed by the depth budget or has no valid split while its centroids differ needs the host's
// nth_element median split: inside k_sah_small it is an insertion sort of the (small) range by one lane; in the large phase the build
// is given back to the caller (HRT_ERR_UNSUPPORTED: host/bvh_build.cpp then builds that mesh itself).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hrt.h"

extern "C" __attribute__((visibility("hidden"))) void hrt_set_last_error(const char* msg);   // hrt_hip.hip

namespace {

#define HRT_SAH_SMALL 64u
#define HRT_SAH_CHUNK 1024u
#define HRT_SAH_BINS 16
#define HRT_SAH_LEVELS 31              // SahBuilder::sahLevels
#define HRT_SAH_NONE 0xffffffffu

struct Ref { float mn[3], mx[3], c[3]; };          // padded triangle box + its centroid (SahBuilder::Ref)
struct Box3 { float mn[3], mx[3]; };

__device__ inline void box_reset(Box3& b) { for (int a = 0; a < 3; ++a) { b.mn[a] = __builtin_huge_valf(); b.mx[a] = -__builtin_huge_valf(); } }
__device__ inline void box_grow(Box3& b, const float* mn, const float* mx) { for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], mn[a]); b.mx[a] = fmaxf(b.mx[a], mx[a]); } }
__device__ inline float half_area(const Box3& b) {                                      // Box3::halfArea
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline Box3 guarded(Box3 x) {                                               // host/bvh_build.cpp refit(): the same arithmetic
    for (int a = 0; a < 3; ++a) {
        const float g = 1e-6f + 4e-7f * fmaxf(fabsf(x.mn[a]), fabsf(x.mx[a]));
        x.mn[a] -= g; x.mx[a] += g;
    }
    return x;
}
__device__ inline int32_t leaf_ref(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }
__device__ inline void set_child(hrt_bvh_node* nodes, uint32_t parent, uint32_t side, const Box3& box, int32_t ref) {
    if (parent == HRT_SAH_NONE) return;
    const Box3 b = guarded(box);
    hrt_bvh_node& n = nodes[parent];
    if (side == 0) { n.c0_min_x = b.mn[0]; n.c0_max_x = b.mx[0]; n.c0_min_y = b.mn[1]; n.c0_max_y = b.mx[1]; n.c0_min_z = b.mn[2]; n.c0_max_z = b.mx[2]; n.child0 = ref; }
    else { n.c1_min_x = b.mn[0]; n.c1_max_x = b.mx[0]; n.c1_min_y = b.mn[1]; n.c1_max_y = b.mx[1]; n.c1_min_z = b.mn[2]; n.c1_max_z = b.mx[2]; n.child1 = ref; }
}
// order-preserving float <-> uint (atomicMin / atomicMax on floats of either sign)
__device__ inline uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float o2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ __launch_bounds__(256) void k_sah_refs(const float* __restrict__ pos, uint32_t n, Ref* __restrict__ refs, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pos + 9ull * i;
    Ref r;
    for (int a = 0; a < 3; ++a) {                       // ITriangle::boundingBox (triangle.cpp:133-151); paddedTriBox of the host
        r.mn[a] = fminf(fminf(p[a], p[3 + a]), p[6 + a]) - 0.0001f;
        r.mx[a] = fmaxf(fmaxf(p[a], p[3 + a]), p[6 + a]) + 0.0001f;
        r.c[a] = 0.5f * (r.mn[a] + r.mx[a]);
    }
    refs[i] = r; idx[i] = i;
}

// a range of the index array whose fate (leaf or inner node) is still open; its result goes into child slot `side` of `parent`
struct Work { uint32_t lo, hi, parent, side; int32_t depth; };
struct WorkStats {      // per large node of the level being run (ordered-uint encoded floats: atomicMin / atomicMax)
    uint32_t box_mn[3], box_mx[3], cb_mn[3], cb_mx[3];
    uint32_t bin_mn[3][HRT_SAH_BINS][3], bin_mx[3][HRT_SAH_BINS][3], bin_cnt[3][HRT_SAH_BINS];
    // the decision of k_sah_eval, for the partition kernels: axis (-1: nothing to move), best bin, centroid minimum and scale on that axis
    int32_t axis, best_bin; float cmin, scale; uint32_t n_left;
};
struct Counters {       // one block of words shared by all kernels
    uint32_t n_nodes;                // allocated so far
    uint32_t n_next_large, n_small;  // work items appended for the next level / for k_sah_small
    uint32_t max_depth;              // inner-node levels
    uint32_t give_up;                // a large node needs the host's median split
    uint32_t n_chunks;               // of the level being run (k_sah_plan)
};

__global__ void k_sah_plan(const Work* __restrict__ work, uint32_t n_work, WorkStats* __restrict__ stats, uint32_t* __restrict__ chunk_work,
                           uint32_t* __restrict__ chunk_start, uint32_t* __restrict__ work_first_chunk, Counters* __restrict__ ctr) {
    // chunk lists: serial prefix by thread 0 (a level has at most n / HRT_SAH_SMALL large nodes), statistics cleared by all
    __shared__ uint32_t total;
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (uint32_t w = 0; w < n_work; ++w) {
            work_first_chunk[w] = c;
            for (uint32_t s = work[w].lo; s < work[w].hi; s += HRT_SAH_CHUNK) { chunk_work[c] = w; chunk_start[c] = s; ++c; }
        }
        work_first_chunk[n_work] = c;
        total = c;
        ctr->n_chunks = c;
    }
    const uint32_t words = sizeof(WorkStats) / 4;
    for (uint32_t i = threadIdx.x; i < n_work * words; i += blockDim.x) {
        const uint32_t k = i % words;
        uint32_t* p = (uint32_t*)&stats[i / words];
        // minima start at the largest ordered value, maxima and counts at 0
        const uint32_t off_mn0 = 0, off_mx0 = 3, off_cmn = 6, off_cmx = 9, off_bmn = 12, off_bmx = 12 + 144, off_cnt = 12 + 288;
        const bool is_min = (k >= off_mn0 && k < off_mx0) || (k >= off_cmn && k < off_cmx) || (k >= off_bmn && k < off_bmx);
        (void)off_cnt;
        p[k] = is_min ? 0xffffffffu : 0u;
    }
}

__device__ inline float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ inline float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__global__ __launch_bounds__(256) void k_sah_bounds(const Work* __restrict__ work, const uint32_t* __restrict__ chunk_work, const uint32_t* __restrict__ chunk_start,
                                                    const Ref* __restrict__ refs, const uint32_t* __restrict__ idx, WorkStats* __restrict__ stats, const Counters* __restrict__ ctr) {
    if (blockIdx.x >= ctr->n_chunks) return;
    const uint32_t w = chunk_work[blockIdx.x], start = chunk_start[blockIdx.x];
    const uint32_t end = min(start + HRT_SAH_CHUNK, work[w].hi);
    Box3 b, c;
    box_reset(b); box_reset(c);
    for (uint32_t i = start + threadIdx.x; i < end; i += blockDim.x) {
        const Ref r = refs[idx[i]];
        box_grow(b, r.mn, r.mx); box_grow(c, r.c, r.c);
    }
    // registers -> wave (shuffles) -> the node's words (one atomic per wave and word; an empty wave's +-inf changes nothing)
    uint32_t* g = (uint32_t*)&stats[w];
    for (int a = 0; a < 3; ++a) {
        const float v0 = wave_min(b.mn[a]), v1 = wave_max(b.mx[a]), v2 = wave_min(c.mn[a]), v3 = wave_max(c.mx[a]);
        if ((threadIdx.x & 63u) == 0 && v0 <= v1) {
            atomicMin(&g[a], f2o(v0)); atomicMax(&g[3 + a], f2o(v1)); atomicMin(&g[6 + a], f2o(v2)); atomicMax(&g[9 + a], f2o(v3));
        }
    }
}

__device__ inline int bin_of(float c, float cmin, float scale) {       // SahBuilder::build: k = (int)((c - cb.mn) * scale), clamped
    int k = (int)((c - cmin) * scale);
    return k < 0 ? 0 : (k >= HRT_SAH_BINS ? HRT_SAH_BINS - 1 : k);
}

__global__ __launch_bounds__(256) void k_sah_bins(const Work* __restrict__ work, const uint32_t* __restrict__ chunk_work, const uint32_t* __restrict__ chunk_start,
                                                  const Ref* __restrict__ refs, const uint32_t* __restrict__ idx, WorkStats* __restrict__ stats, const Counters* __restrict__ ctr) {
    __shared__ uint32_t s_mn[3][HRT_SAH_BINS][3], s_mx[3][HRT_SAH_BINS][3], s_cnt[3][HRT_SAH_BINS];
    if (blockIdx.x >= ctr->n_chunks) return;
    for (uint32_t i = threadIdx.x; i < 3 * HRT_SAH_BINS * 3; i += blockDim.x) { ((uint32_t*)s_mn)[i] = 0xffffffffu; ((uint32_t*)s_mx)[i] = 0u; }
    for (uint32_t i = threadIdx.x; i < 3 * HRT_SAH_BINS; i += blockDim.x) ((uint32_t*)s_cnt)[i] = 0u;
    __syncthreads();
    const uint32_t w = chunk_work[blockIdx.x], start = chunk_start[blockIdx.x];
    const uint32_t end = min(start + HRT_SAH_CHUNK, work[w].hi);
    float cmin[3], scale[3]; bool use[3];
    for (int a = 0; a < 3; ++a) {
        cmin[a] = o2f(stats[w].cb_mn[a]);
        const float e = o2f(stats[w].cb_mx[a]) - cmin[a];
        use[a] = e > 0;
        scale[a] = HRT_SAH_BINS / e;
    }
    for (uint32_t i = start + threadIdx.x; i < end; i += blockDim.x) {
        const Ref r = refs[idx[i]];
        for (int a = 0; a < 3; ++a) {
            if (!use[a]) continue;
            const int k = bin_of(r.c[a], cmin[a], scale[a]);
            for (int d = 0; d < 3; ++d) { atomicMin(&s_mn[a][k][d], f2o(r.mn[d])); atomicMax(&s_mx[a][k][d], f2o(r.mx[d])); }
            atomicAdd(&s_cnt[a][k], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3 * HRT_SAH_BINS; i += blockDim.x) {
        const uint32_t c = ((uint32_t*)s_cnt)[i];
        if (!c) continue;
        atomicAdd(&((uint32_t*)stats[w].bin_cnt)[i], c);
        for (int d = 0; d < 3; ++d) {
            atomicMin(&((uint32_t*)stats[w].bin_mn)[3 * i + d], ((uint32_t*)s_mn)[3 * i + d]);
            atomicMax(&((uint32_t*)stats[w].bin_mx)[3 * i + d], ((uint32_t*)s_mx)[3 * i + d]);
        }
    }
}

// The sweep of SahBuilder::build over one axis' bins: best (cost, bin) with "first strictly better wins" carried in and out.
__device__ inline void sweep_axis(int a, const Box3* bb, const uint32_t* bc, float& bestCost, int& bestAxis, int& bestBin) {
    float rightArea[HRT_SAH_BINS]; uint32_t rightCount[HRT_SAH_BINS];
    Box3 acc; box_reset(acc); uint32_t cnt = 0;
    for (int k = HRT_SAH_BINS - 1; k > 0; --k) { box_grow(acc, bb[k].mn, bb[k].mx); cnt += bc[k]; rightArea[k] = half_area(acc); rightCount[k] = cnt; }
    box_reset(acc); cnt = 0;
    for (int k = 0; k < HRT_SAH_BINS - 1; ++k) {
        box_grow(acc, bb[k].mn, bb[k].mx); cnt += bc[k];
        if (cnt == 0 || rightCount[k + 1] == 0) continue;
        const float cost = half_area(acc) * cnt + rightArea[k + 1] * rightCount[k + 1];
        if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
    }
}
__device__ inline int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

// appends a child range: to k_sah_small's list or to the next level's
__device__ inline void push_child(Work* next_large, Work* small, Counters* ctr, uint32_t lo, uint32_t hi, uint32_t parent, uint32_t side, int depth) {
    Work c; c.lo = lo; c.hi = hi; c.parent = parent; c.side = side; c.depth = depth;
    if (hi - lo > HRT_SAH_SMALL) next_large[atomicAdd(&ctr->n_next_large, 1u)] = c;
    else small[atomicAdd(&ctr->n_small, 1u)] = c;
}

__global__ __launch_bounds__(64) void k_sah_eval(const Work* __restrict__ work, uint32_t n_work, WorkStats* __restrict__ stats, uint32_t max_leaf,
                                                 hrt_bvh_node* __restrict__ nodes, Work* __restrict__ next_large, Work* __restrict__ small, Counters* __restrict__ ctr) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_work) return;
    const Work wk = work[w];
    WorkStats& st = stats[w];
    const uint32_t n = wk.hi - wk.lo;                   // > HRT_SAH_SMALL >= max_leaf: never a leaf
    Box3 outBox, cb;
    for (int a = 0; a < 3; ++a) { outBox.mn[a] = o2f(st.box_mn[a]); outBox.mx[a] = o2f(st.box_mx[a]); cb.mn[a] = o2f(st.cb_mn[a]); cb.mx[a] = o2f(st.cb_mx[a]); }
    const bool sahAllowed = wk.depth + ceil_log2(n) + 1 < HRT_SAH_LEVELS;
    float bestCost = __builtin_huge_valf(); int bestAxis = -1, bestBin = -1;
    if (sahAllowed) {
        for (int a = 0; a < 3; ++a) {
            const float e = cb.mx[a] - cb.mn[a];
            if (!(e > 0)) continue;
            Box3 bb[HRT_SAH_BINS]; uint32_t bc[HRT_SAH_BINS];
            for (int k = 0; k < HRT_SAH_BINS; ++k) {
                bc[k] = st.bin_cnt[a][k];
                if (bc[k]) for (int d = 0; d < 3; ++d) { bb[k].mn[d] = o2f(st.bin_mn[a][k][d]); bb[k].mx[d] = o2f(st.bin_mx[a][k][d]); }
                else box_reset(bb[k]);
            }
            sweep_axis(a, bb, bc, bestCost, bestAxis, bestBin);
        }
    }
    uint32_t mid;
    st.axis = -1; st.best_bin = 0; st.cmin = 0.0f; st.scale = 0.0f; st.n_left = 0;
    if (bestAxis >= 0) {
        uint32_t nl = 0;
        for (int k = 0; k <= bestBin; ++k) nl += st.bin_cnt[bestAxis][k];
        mid = wk.lo + nl;                               // 0 < nl < n: the sweep only prices splits with both sides occupied
        st.axis = bestAxis; st.best_bin = bestBin; st.cmin = cb.mn[bestAxis]; st.scale = HRT_SAH_BINS / (cb.mx[bestAxis] - cb.mn[bestAxis]); st.n_left = nl;
    } else if (cb.mx[0] == cb.mn[0] && cb.mx[1] == cb.mn[1] && cb.mx[2] == cb.mn[2]) {
        mid = wk.lo + n / 2;                            // every centroid in one point: the median split needs no reordering
    } else {
        atomicExch(&ctr->give_up, 1u);                  // the host's nth_element median split of a large range: not here
        return;
    }
    const uint32_t me = atomicAdd(&ctr->n_nodes, 1u);
    atomicMax(&ctr->max_depth, (uint32_t)wk.depth);
    set_child(nodes, wk.parent, wk.side, outBox, (int32_t)me);
    push_child(next_large, small, ctr, wk.lo, mid, me, 0, wk.depth + 1);
    push_child(next_large, small, ctr, mid, wk.hi, me, 1, wk.depth + 1);
}

// partition of a large node's range: count the "left" refs of every chunk, then scatter
__global__ __launch_bounds__(256) void k_sah_count(const Work* __restrict__ work, const uint32_t* __restrict__ chunk_work, const uint32_t* __restrict__ chunk_start,
                                                   const Ref* __restrict__ refs, const uint32_t* __restrict__ idx, const WorkStats* __restrict__ stats,
                                                   uint32_t* __restrict__ chunk_left, const Counters* __restrict__ ctr) {
    __shared__ uint32_t s;
    if (blockIdx.x >= ctr->n_chunks) return;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const uint32_t w = chunk_work[blockIdx.x], start = chunk_start[blockIdx.x];
    const uint32_t end = min(start + HRT_SAH_CHUNK, work[w].hi);
    const int axis = stats[w].axis;
    uint32_t mine = 0;
    if (axis >= 0) {
        const float cmin = stats[w].cmin, scale = stats[w].scale; const int bb = stats[w].best_bin;
        for (uint32_t i = start + threadIdx.x; i < end; i += blockDim.x) mine += bin_of(refs[idx[i]].c[axis], cmin, scale) <= bb ? 1u : 0u;
    }
    if (mine) atomicAdd(&s, mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_left[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sah_scatter(const Work* __restrict__ work, const uint32_t* __restrict__ chunk_work, const uint32_t* __restrict__ chunk_start,
                                                     const uint32_t* __restrict__ work_first_chunk, const Ref* __restrict__ refs, const uint32_t* __restrict__ idx_in,
                                                     uint32_t* __restrict__ idx_out, const WorkStats* __restrict__ stats, const uint32_t* __restrict__ chunk_left,
                                                     const Counters* __restrict__ ctr) {
    __shared__ uint32_t s_left, s_right;
    if (blockIdx.x >= ctr->n_chunks) return;
    const uint32_t w = chunk_work[blockIdx.x], start = chunk_start[blockIdx.x];
    const uint32_t end = min(start + HRT_SAH_CHUNK, work[w].hi);
    const int axis = stats[w].axis;
    if (axis < 0) {      // nothing moves
        for (uint32_t i = start + threadIdx.x; i < end; i += blockDim.x) idx_out[i] = idx_in[i];
        return;
    }
    if (threadIdx.x == 0) {
        uint32_t l = 0, r = 0;
        for (uint32_t c = work_first_chunk[w]; c < blockIdx.x; ++c) { l += chunk_left[c]; r += min(HRT_SAH_CHUNK, work[w].hi - chunk_start[c]) - chunk_left[c]; }
        s_left = work[w].lo + l; s_right = work[w].lo + stats[w].n_left + r;
    }
    __syncthreads();
    // A STABLE partition (ranks from ballots and a prefix over the block's four waves, 256 positions per round): the leaf order, and
    // with it the reference tree restated over the soup in leaf order (host/bvh_build.cpp referenceLeafBoxes), must not depend on
    // which thread came first -- two loads of one file (two ranks of a multi-GPU job) must flatten to the same scene.
    __shared__ uint32_t s_wl[4], s_wr[4];
    const float cmin = stats[w].cmin, scale = stats[w].scale; const int bb = stats[w].best_bin;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t i0 = start; i0 < end; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const bool live = i < end;
        uint32_t t = 0; bool left = false;
        if (live) { t = idx_in[i]; left = bin_of(refs[t].c[axis], cmin, scale) <= bb; }
        const unsigned long long ml = __ballot(live && left), mr = __ballot(live && !left);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { s_wl[wv] = (uint32_t)__popcll(ml); s_wr[wv] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t bl = 0, br = 0, tl = 0, tr = 0;
        for (uint32_t k = 0; k < 4; ++k) { if (k < wv) { bl += s_wl[k]; br += s_wr[k]; } tl += s_wl[k]; tr += s_wr[k]; }
        if (live) idx_out[left ? s_left + bl + (uint32_t)__popcll(ml & below) : s_right + br + (uint32_t)__popcll(mr & below)] = t;
        __syncthreads();
        if (threadIdx.x == 0) { s_left += tl; s_right += tr; }
        __syncthreads();
    }
}

// One WAVE per small subtree: SahBuilder::build with an explicit stack, the 64 lanes sharing each node's passes over its
// triangles (HRT_SAH_SMALL / 64 per lane; the partition below holds two per lane in registers: HRT_SAH_SMALL <= 128).  The subtree's refs are copied into LDS once; `lidx` is the wave's private index array
// (positions into that copy), permuted in place and written back to idx[lo, hi) at the end.
#define HRT_SAH_WAVES 4
struct SmallLds {
    float ref[9][HRT_SAH_SMALL];                       // mn.xyz, mx.xyz, c.xyz of the subtree's triangles (SoA)
    uint32_t gidx[HRT_SAH_SMALL];                      // their indices in the mesh
    uint32_t lidx[HRT_SAH_SMALL];
    uint32_t bin_mn[3][HRT_SAH_BINS][3], bin_mx[3][HRT_SAH_BINS][3], bin_cnt[3][HRT_SAH_BINS];
    Work stack[40];
};
__device__ inline unsigned lanes_below_mask(unsigned long long m, unsigned lane) { return (unsigned)__popcll(m & ((1ull << lane) - 1ull)); }
__global__ __launch_bounds__(64 * HRT_SAH_WAVES) void k_sah_small(const Work* __restrict__ small, uint32_t n_small, const Ref* __restrict__ refs, uint32_t* __restrict__ idx,
                                                                 uint32_t max_leaf, hrt_bvh_node* __restrict__ nodes, Counters* __restrict__ ctr) {
    __shared__ SmallLds lds_all[HRT_SAH_WAVES];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t s = blockIdx.x * HRT_SAH_WAVES + wv;
    if (s >= n_small) return;                           // (whole waves leave: no block-wide barrier below)
    SmallLds& L = lds_all[wv];
    const float triCost = 1.3f, boxCost = 1.0f;
    const Work root = small[s];
    const uint32_t base = root.lo, total = root.hi - root.lo;       // <= HRT_SAH_SMALL
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t t = idx[base + i];
        const Ref r = refs[t];
        L.gidx[i] = t; L.lidx[i] = i;
        for (int a = 0; a < 3; ++a) { L.ref[a][i] = r.mn[a]; L.ref[3 + a][i] = r.mx[a]; L.ref[6 + a][i] = r.c[a]; }
    }
    int sp = 0;
    if (lane == 0) { Work r0 = root; r0.lo = 0; r0.hi = total; L.stack[0] = r0; }
    sp = 1;
    uint32_t my_max_depth = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (sp > 0) {
        --sp;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const Work wk = L.stack[sp];                    // (the same word in every lane)
        const uint32_t lo = wk.lo, hi = wk.hi, n = hi - lo;
        // ---- bounds (each lane: positions lo + lane and lo + lane + 64)
        Box3 outBox, cb;
        box_reset(outBox); box_reset(cb);
        for (uint32_t p = lo + lane; p < hi; p += 64) {
            const uint32_t k = L.lidx[p];
            float mn[3] = {L.ref[0][k], L.ref[1][k], L.ref[2][k]}, mx[3] = {L.ref[3][k], L.ref[4][k], L.ref[5][k]}, c[3] = {L.ref[6][k], L.ref[7][k], L.ref[8][k]};
            box_grow(outBox, mn, mx); box_grow(cb, c, c);
        }
        for (int a = 0; a < 3; ++a) { outBox.mn[a] = wave_min(outBox.mn[a]); outBox.mx[a] = wave_max(outBox.mx[a]); cb.mn[a] = wave_min(cb.mn[a]); cb.mx[a] = wave_max(cb.mx[a]); }
        if (n == 1) { if (lane == 0) set_child(nodes, wk.parent, wk.side, outBox, leaf_ref(base + lo, 1)); continue; }
        uint32_t mid = lo;
        bool haveSplit = false, leaf = false;
        int axis = 0;
        { float ext = -1; for (int a = 0; a < 3; ++a) { const float e = cb.mx[a] - cb.mn[a]; if (e > ext) { ext = e; axis = a; } } }
        const float leafCost = triCost * n;
        const bool sahAllowed = wk.depth + ceil_log2(n) + 1 < HRT_SAH_LEVELS;
        if (sahAllowed && n > 2) {
            // ---- bins of the three axes in one pass (LDS atomics on ordered integers)
            for (uint32_t i = lane; i < 3 * HRT_SAH_BINS * 3; i += 64) { ((uint32_t*)L.bin_mn)[i] = 0xffffffffu; ((uint32_t*)L.bin_mx)[i] = 0u; }
            if (lane < 3 * HRT_SAH_BINS) ((uint32_t*)L.bin_cnt)[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float scale[3]; bool use[3];
            for (int a = 0; a < 3; ++a) { const float e = cb.mx[a] - cb.mn[a]; use[a] = e > 0; scale[a] = HRT_SAH_BINS / e; }
            for (uint32_t p = lo + lane; p < hi; p += 64) {
                const uint32_t k = L.lidx[p];
                for (int a = 0; a < 3; ++a) {
                    if (!use[a]) continue;
                    const int bk = bin_of(L.ref[6 + a][k], cb.mn[a], scale[a]);
                    for (int d = 0; d < 3; ++d) { atomicMin(&L.bin_mn[a][bk][d], f2o(L.ref[d][k])); atomicMax(&L.bin_mx[a][bk][d], f2o(L.ref[3 + d][k])); }
                    atomicAdd(&L.bin_cnt[a][bk], 1u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // ---- the sweep: lane a prices axis a; combined in axis order with "first strictly better wins"
            float myCost = __builtin_huge_valf(); int myAxis = -1, myBin = -1;
            if (lane < 3 && use[lane]) {
                Box3 bb[HRT_SAH_BINS]; uint32_t bc[HRT_SAH_BINS];
                for (int k = 0; k < HRT_SAH_BINS; ++k) {
                    bc[k] = L.bin_cnt[lane][k];
                    if (bc[k]) for (int d = 0; d < 3; ++d) { bb[k].mn[d] = o2f(L.bin_mn[lane][k][d]); bb[k].mx[d] = o2f(L.bin_mx[lane][k][d]); }
                    else box_reset(bb[k]);
                }
                sweep_axis((int)lane, bb, bc, myCost, myAxis, myBin);
            }
            float bestCost = __builtin_huge_valf(); int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                const float c = __shfl(myCost, a, 64); const int ax = __shfl(myAxis, a, 64), bn = __shfl(myBin, a, 64);
                if (ax >= 0 && c < bestCost) { bestCost = c; bestAxis = ax; bestBin = bn; }
            }
            if (bestAxis >= 0) {
                const float parentArea = half_area(outBox);
                const float splitCost = 2 * boxCost + triCost * bestCost / (parentArea > 0 ? parentArea : 1.0f);
                if (n <= max_leaf && leafCost <= splitCost) leaf = true;
                else {
                    // ---- stable partition by "bin <= bestBin": ranks from ballots, two positions per lane, values held in registers
                    const float sc = HRT_SAH_BINS / (cb.mx[bestAxis] - cb.mn[bestAxis]);
                    uint32_t v[2] = {0, 0}; bool live[2], left[2];
                    unsigned long long ml[2], mr[2];
                    for (int r = 0; r < 2; ++r) {
                        const uint32_t p = lo + (uint32_t)r * 64u + lane;
                        live[r] = p < hi; left[r] = false;
                        if (live[r]) { v[r] = L.lidx[p]; left[r] = bin_of(L.ref[6 + bestAxis][v[r]], cb.mn[bestAxis], sc) <= bestBin; }
                        ml[r] = __ballot(live[r] && left[r]); mr[r] = __ballot(live[r] && !left[r]);
                    }
                    const uint32_t nl = (uint32_t)(__popcll(ml[0]) + __popcll(ml[1]));
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (int r = 0; r < 2; ++r) {
                        if (!live[r]) continue;
                        const uint32_t before_l = (r ? (uint32_t)__popcll(ml[0]) : 0u) + lanes_below_mask(ml[r], lane);
                        const uint32_t before_r = (r ? (uint32_t)__popcll(mr[0]) : 0u) + lanes_below_mask(mr[r], lane);
                        L.lidx[left[r] ? lo + before_l : lo + nl + before_r] = v[r];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    mid = lo + nl;
                    haveSplit = mid > lo && mid < hi;
                }
            }
        }
        if (!leaf && !haveSplit) {
            if (n <= max_leaf) leaf = true;
            else {
                // balanced median split on the widest centroid axis (std::nth_element on the host): a small range, sorted by lane 0
                mid = lo + n / 2;
                if (lane == 0) {
                    for (uint32_t i = lo + 1; i < hi; ++i) {
                        const uint32_t t = L.lidx[i]; const float key = L.ref[6 + axis][t];
                        uint32_t j = i;
                        while (j > lo && L.ref[6 + axis][L.lidx[j - 1]] > key) { L.lidx[j] = L.lidx[j - 1]; --j; }
                        L.lidx[j] = t;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        if (leaf) { if (lane == 0) set_child(nodes, wk.parent, wk.side, outBox, leaf_ref(base + lo, n)); continue; }
        uint32_t me = 0;
        if (lane == 0) me = atomicAdd(&ctr->n_nodes, 1u);
        me = (uint32_t)__shfl((int)me, 0, 64);
        if ((uint32_t)wk.depth > my_max_depth) my_max_depth = (uint32_t)wk.depth;
        if (sp + 2 > 40) { if (lane == 0) atomicExch(&ctr->give_up, 2u); return; }      // (cannot happen: the depth budget bounds the stack at 32)
        if (lane == 0) {
            set_child(nodes, wk.parent, wk.side, outBox, (int32_t)me);
            Work c; c.parent = me; c.depth = wk.depth + 1;
            c.lo = mid; c.hi = hi; c.side = 1; L.stack[sp] = c;
            c.lo = lo; c.hi = mid; c.side = 0; L.stack[sp + 1] = c;
        }
        sp += 2;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t i = lane; i < total; i += 64) idx[base + i] = L.gidx[L.lidx[i]];
    if (lane == 0 && my_max_depth) atomicMax(&ctr->max_depth, my_max_depth);
}

struct DevBufs {
    void* p[40]; int n = 0;
    ~DevBufs() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
    template <class T> hipError_t get(T** out, size_t count) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e == hipSuccess) { p[n++] = q; *out = (T*)q; }
        return e;
    }
};
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
struct OwnStream {
    hipStream_t s = nullptr;
    ~OwnStream() { if (s) (void)hipStreamDestroy(s); }
};
hrt_status sfail(hrt_status st, const std::string& msg) { hrt_set_last_error(msg.c_str()); return st; }
#define SCHK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return sfail(e_ == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define SLAUNCH(name) do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return sfail(HRT_ERR_HIP, std::string(name " launch: ") + hipGetErrorString(e_)); } while (0)

hrt_status build_sah_impl(int device, const float* tri_pos, uint32_t n, uint32_t max_leaf, hrt_bvh_node* nodes_out, uint32_t* n_nodes_out,
                          uint32_t* order_out, int32_t* depth_out) {
    if (!tri_pos || !nodes_out || !n_nodes_out || !order_out || !depth_out) return sfail(HRT_ERR_INVALID, "hrt_bvh_build_sah: NULL argument");
    if (max_leaf < 1 || max_leaf > 8) return sfail(HRT_ERR_INVALID, "hrt_bvh_build_sah: max_leaf must be 1..8");
    if (n <= max_leaf || n >= (1u << 28)) return sfail(HRT_ERR_INVALID, "hrt_bvh_build_sah: needs max_leaf < n_tris < 2^28");
    for (uint64_t k = 0; k < 9ull * n; ++k)
        if (!(tri_pos[k] - tri_pos[k] == 0.0f)) return sfail(HRT_ERR_INVALID, "hrt_bvh_build_sah: non-finite vertex");
    int n_dev = 0;
    SCHK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return sfail(HRT_ERR_NO_DEVICE, "hrt_bvh_build_sah: no such device");
    DeviceGuard guard;
    SCHK(hipSetDevice(device));
    OwnStream own;
    SCHK(hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking));
    hipStream_t stream = own.s;

    const size_t max_large = (size_t)n / HRT_SAH_SMALL + 2;              // large nodes of one level (disjoint ranges of more than SMALL triangles)
    const size_t max_chunks = (size_t)n / HRT_SAH_CHUNK + max_large + 2;
    const size_t max_small = (size_t)n + 2;                               // (every small range holds at least one triangle)
    // one allocation, carved up (fifteen hipMalloc / hipFree pairs were 2 ms of a 100 k-triangle build)
    DevBufs bufs;
    float* d_pos; Ref* d_refs; uint32_t *d_idx[2], *d_chunk_work, *d_chunk_start, *d_first_chunk, *d_chunk_left;
    Work *d_work[2], *d_small; WorkStats* d_stats; Counters* d_ctr; hrt_bvh_node* d_nodes;
    {
        size_t off = 0;
        auto place = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
        const size_t o_pos = place(9ull * n * sizeof(float)), o_refs = place((size_t)n * sizeof(Ref)), o_i0 = place((size_t)n * 4), o_i1 = place((size_t)n * 4),
                     o_cw = place(max_chunks * 4), o_cs = place(max_chunks * 4), o_cl = place(max_chunks * 4), o_fc = place((max_large + 1) * 4),
                     o_w0 = place(max_large * sizeof(Work)), o_w1 = place(max_large * sizeof(Work)), o_sm = place(max_small * sizeof(Work)),
                     o_st = place(max_large * sizeof(WorkStats)), o_ct = place(sizeof(Counters)), o_nd = place((size_t)n * sizeof(hrt_bvh_node));
        char* base = nullptr;
        SCHK(bufs.get(&base, off));
        d_pos = (float*)(base + o_pos); d_refs = (Ref*)(base + o_refs); d_idx[0] = (uint32_t*)(base + o_i0); d_idx[1] = (uint32_t*)(base + o_i1);
        d_chunk_work = (uint32_t*)(base + o_cw); d_chunk_start = (uint32_t*)(base + o_cs); d_chunk_left = (uint32_t*)(base + o_cl); d_first_chunk = (uint32_t*)(base + o_fc);
        d_work[0] = (Work*)(base + o_w0); d_work[1] = (Work*)(base + o_w1); d_small = (Work*)(base + o_sm); d_stats = (WorkStats*)(base + o_st);
        d_ctr = (Counters*)(base + o_ct); d_nodes = (hrt_bvh_node*)(base + o_nd);
    }
    SCHK(hipMemcpyAsync(d_pos, tri_pos, 9ull * n * sizeof(float), hipMemcpyHostToDevice, stream));
    SCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), stream));
    SCHK(hipMemsetAsync(d_nodes, 0, (size_t)n * sizeof(hrt_bvh_node), stream));
    hipLaunchKernelGGL(k_sah_refs, dim3((n + 255) / 256), dim3(256), 0, stream, d_pos, n, d_refs, d_idx[0]);
    SLAUNCH("k_sah_refs");

    Work root; root.lo = 0; root.hi = n; root.parent = HRT_SAH_NONE; root.side = 0; root.depth = 1;
    int cur = 0;           // which index array holds the current order
    uint32_t n_work = 0;
    if (n > HRT_SAH_SMALL) { SCHK(hipMemcpyAsync(d_work[0], &root, sizeof(root), hipMemcpyHostToDevice, stream)); n_work = 1; }
    else {
        SCHK(hipMemcpyAsync(d_small, &root, sizeof(root), hipMemcpyHostToDevice, stream));
        const uint32_t one = 1;
        SCHK(hipMemcpyAsync(&d_ctr->n_small, &one, 4, hipMemcpyHostToDevice, stream));
    }
    int wcur = 0;
    Counters h{};
    for (int level = 0; n_work > 0; ++level) {
        if (level > 64) return sfail(HRT_ERR_HIP, "hrt_bvh_build_sah: the large-node phase does not end");
        hipLaunchKernelGGL(k_sah_plan, dim3(1), dim3(256), 0, stream, d_work[wcur], n_work, d_stats, d_chunk_work, d_chunk_start, d_first_chunk, d_ctr);
        SLAUNCH("k_sah_plan");
        // the chunk count stays on the device: the ranges are disjoint, so n / CHUNK + n_work bounds it, and the blocks beyond
        // ctr->n_chunks leave at once (one host synchronisation per level instead of two)
        const uint32_t n_chunks = (uint32_t)std::min<size_t>(max_chunks, (size_t)n / HRT_SAH_CHUNK + n_work);
        hipLaunchKernelGGL(k_sah_bounds, dim3(n_chunks), dim3(256), 0, stream, d_work[wcur], d_chunk_work, d_chunk_start, d_refs, d_idx[cur], d_stats, d_ctr);
        hipLaunchKernelGGL(k_sah_bins, dim3(n_chunks), dim3(256), 0, stream, d_work[wcur], d_chunk_work, d_chunk_start, d_refs, d_idx[cur], d_stats, d_ctr);
        SLAUNCH("k_sah_bounds / k_sah_bins");
        hipLaunchKernelGGL(k_sah_eval, dim3((n_work + 63) / 64), dim3(64), 0, stream, d_work[wcur], n_work, d_stats, max_leaf, d_nodes, d_work[wcur ^ 1], d_small, d_ctr);
        hipLaunchKernelGGL(k_sah_count, dim3(n_chunks), dim3(256), 0, stream, d_work[wcur], d_chunk_work, d_chunk_start, d_refs, d_idx[cur], d_stats, d_chunk_left, d_ctr);
        SLAUNCH("k_sah_eval / k_sah_count");
        // ranges of this level's large nodes are rewritten into the other index array; everything else is copied first
        SCHK(hipMemcpyAsync(d_idx[cur ^ 1], d_idx[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_sah_scatter, dim3(n_chunks), dim3(256), 0, stream, d_work[wcur], d_chunk_work, d_chunk_start, d_first_chunk, d_refs, d_idx[cur], d_idx[cur ^ 1],
                           d_stats, d_chunk_left, d_ctr);
        SLAUNCH("k_sah_scatter");
        cur ^= 1;
        SCHK(hipMemcpyAsync(&h, d_ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        SCHK(hipStreamSynchronize(stream));
        if (h.give_up) return sfail(HRT_ERR_UNSUPPORTED, "hrt_bvh_build_sah: a large node needs the median split of the host builder (depth budget or no SAH split): build this mesh on the host");
        if (h.n_next_large > max_large || h.n_small > max_small) return sfail(HRT_ERR_HIP, "hrt_bvh_build_sah: work list overflow");
        n_work = h.n_next_large;
        const uint32_t zero = 0;
        SCHK(hipMemcpyAsync(&d_ctr->n_next_large, &zero, 4, hipMemcpyHostToDevice, stream));
        wcur ^= 1;
    }
    SCHK(hipMemcpyAsync(&h, d_ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    SCHK(hipStreamSynchronize(stream));
    if (h.n_small) {
        hipLaunchKernelGGL(k_sah_small, dim3((h.n_small + HRT_SAH_WAVES - 1) / HRT_SAH_WAVES), dim3(64 * HRT_SAH_WAVES), 0, stream, d_small, h.n_small, d_refs, d_idx[cur],
                           max_leaf, d_nodes, d_ctr);
        SLAUNCH("k_sah_small");
    }
    SCHK(hipMemcpyAsync(&h, d_ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
    SCHK(hipStreamSynchronize(stream));
    if (h.give_up) return sfail(HRT_ERR_HIP, "hrt_bvh_build_sah: traversal stack of the small-subtree kernel exhausted");
    if (h.n_nodes == 0 || h.n_nodes > n - 1) return sfail(HRT_ERR_HIP, "hrt_bvh_build_sah: bad node count");
    SCHK(hipMemcpy(nodes_out, d_nodes, (size_t)h.n_nodes * sizeof(hrt_bvh_node), hipMemcpyDeviceToHost));
    SCHK(hipMemcpy(order_out, d_idx[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_nodes_out = h.n_nodes;
    *depth_out = (int32_t)h.max_depth;
    return HRT_OK;
}

}  // namespace

extern "C" hrt_status hrt_bvh_build_sah(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                                        uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out) {
    try {
        return build_sah_impl(device, tri_pos, n_tris, max_leaf, nodes_out, n_nodes_out, order_out, depth_out);
    } catch (const std::bad_alloc&) { return sfail(HRT_ERR_OOM, "hrt_bvh_build_sah: out of host memory"); }
    catch (const std::exception& e) { return sfail(HRT_ERR_INVALID, std::string("hrt_bvh_build_sah: ") + e.what()); }
    catch (...) { return sfail(HRT_ERR_INVALID, "hrt_bvh_build_sah: unknown C++ exception"); }
}
