// hrt_lbvh.hip -- hrt_bvh_build_device: the culling tree of a mesh built on the GPU.
//
// What it replaces: the BVHNode constructor (bvh.cpp:6-61) as far as the TOPOLOGY of the flattened tree goes -- the same
// job host/bvh_build.cpp's binned-SAH builder does on the CPU.  The closest hit does not depend on that topology (DESIGN.md
// section 2: the reference's own tree is restated separately, over the soup in leaf order), so the tree may be any tree;
// this one is a Morton-ordered LBVH (Karras 2012): fast to build (one radix sort + three passes over the triangles), ~20 %
// more node visits per ray than the SAH tree.  For meshes whose build time matters more than a render's last 20 %.
//
//   1. k_prep      per triangle: ITriangle::boundingBox (triangle.cpp:133-151: +-1e-4 padding), centroid of that box;
//                  per block a reduction of the centroid bounds, folded with ordered-integer atomics
//   2. k_codes     63-bit Morton code of the centroid on a 2^21 grid of those bounds
//   3. rocprim::radix_sort_pairs (code, triangle)     -- a library sort: nothing here is worth a hand-written one
//   4. k_hierarchy one thread per internal node: its range and split from the common prefixes of the sorted codes
//                  (ties broken by position, so equal codes still give a balanced subtree)
//   5. k_flag + rocprim::exclusive_scan: nodes that span at most `max_leaf` triangles become leaves (the sorted order
//                  keeps a node's triangles contiguous); the others are renumbered densely, root = 0
//   6. k_refit     bottom-up, one thread per triangle, the second thread to reach a node goes on (atomic flag): box of a
//                  node = union of the padded triangle boxes below it; depth of the tree as the traversal's stack sees it
//   7. k_emit      hrt_bvh_node records: child boxes widened by the same rounding guard host/bvh_build.cpp refit() applies
//                  (the kernel's fma slab test must never be tighter than a division-based test of the same box)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/hrt.h"

extern "C" __attribute__((visibility("hidden"))) void hrt_set_last_error(const char* msg);   // hrt_hip.hip

namespace {

struct Box { float mn[3], mx[3]; };

__device__ inline float fminf3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ inline float fmaxf3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
__device__ inline uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float o2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ __launch_bounds__(256) void k_prep(const float* __restrict__ pos, uint32_t n, Box* __restrict__ tbox, uint32_t* __restrict__ bounds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    const bool live = i < n;
    if (live) {
        const float* p = pos + 9ull * i;
        Box b;
        for (int a = 0; a < 3; ++a) {                       // glm::min / glm::max of triangle.cpp:135-146 on finite input
            b.mn[a] = fminf3(p[a], p[3 + a], p[6 + a]) - 0.0001f;
            b.mx[a] = fmaxf3(p[a], p[3 + a], p[6 + a]) + 0.0001f;
            c[a] = 0.5f * (b.mn[a] + b.mx[a]);
        }
        tbox[i] = b;
    }
    __shared__ uint32_t s_lo[3], s_hi[3];
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0xffffffffu; s_hi[threadIdx.x] = 0u; }
    __syncthreads();
    if (live)
        for (int a = 0; a < 3; ++a)
            if (c[a] == c[a]) { atomicMin(&s_lo[a], f2o(c[a])); atomicMax(&s_hi[a], f2o(c[a])); }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&bounds[threadIdx.x], s_lo[threadIdx.x]); atomicMax(&bounds[3 + threadIdx.x], s_hi[threadIdx.x]); }
}

__device__ inline uint64_t spread21(uint64_t x) {           // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(256) void k_codes(const Box* __restrict__ tbox, uint32_t n, const uint32_t* __restrict__ bounds,
                                               uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = o2f(bounds[a]), hi = o2f(bounds[3 + a]);
        const float c = 0.5f * (tbox[i].mn[a] + tbox[i].mx[a]);
        // (in double: hi - lo may overflow fp32 for a mesh as wide as fp32 allows; NaN centroids go to cell 0)
        const double e = (double)hi - (double)lo;
        double f = e > 0 ? ((double)c - (double)lo) / e : 0.0;
        f = f == f ? f : 0.0;
        f = f < 0 ? 0 : (f > 1 ? 1 : f);
        uint64_t q = (uint64_t)(f * 2097151.0);
        code |= spread21(q) << (2 - a);
    }
    keys[i] = code; vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (Karras 2012, section 4), -1 outside the array;
// equal keys: the prefix goes on into the positions themselves
__device__ inline int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}
// Internal node i in [0, n - 2]: range, split, children.  A child is internal node `gamma` (or gamma + 1) or the single
// triangle at that sorted position; kid = index | 0x80000000 for a triangle.
__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t* __restrict__ keys, int n, uint2* __restrict__ range, uint2* __restrict__ kids,
                                                   uint32_t* __restrict__ parent_of_inner, uint32_t* __restrict__ parent_of_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
    uint2 k;
    if (lo == gamma) { k.x = (uint32_t)gamma | 0x80000000u; parent_of_leaf[gamma] = (uint32_t)i; }
    else { k.x = (uint32_t)gamma; parent_of_inner[gamma] = (uint32_t)i; }
    if (hi == gamma + 1) { k.y = (uint32_t)(gamma + 1) | 0x80000000u; parent_of_leaf[gamma + 1] = (uint32_t)i; }
    else { k.y = (uint32_t)(gamma + 1); parent_of_inner[gamma + 1] = (uint32_t)i; }
    kids[i] = k;
    if (i == 0) parent_of_inner[0] = 0xffffffffu;
}

__global__ __launch_bounds__(256) void k_flag(const uint2* __restrict__ range, int n_inner, uint32_t max_leaf, uint32_t* __restrict__ real) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) real[i] = (range[i].y - range[i].x + 1u > max_leaf) ? 1u : 0u;
}

// Bottom-up: boxes and stack depth.  nbox[0 .. n_inner) internal nodes; depth = inner levels below and including the node,
// counting only nodes that stay inner nodes.
__global__ __launch_bounds__(256) void k_refit(const Box* __restrict__ tbox, const uint32_t* __restrict__ vals, int n, const uint2* __restrict__ kids,
                                               const uint32_t* __restrict__ parent_of_inner, const uint32_t* __restrict__ parent_of_leaf,
                                               const uint32_t* __restrict__ real, Box* __restrict__ nbox, uint32_t* __restrict__ ndepth,
                                               uint32_t* __restrict__ visits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parent_of_leaf[i];
    while (node != 0xffffffffu) {
        __threadfence();
        if (atomicAdd(&visits[node], 1u) == 0u) return;          // the first to arrive leaves the node to the second
        __threadfence();
        const uint2 k = kids[node];
        Box b;
        uint32_t dep = 0;
        for (int c = 0; c < 2; ++c) {
            const uint32_t kid = c == 0 ? k.x : k.y;
            Box cb;
            if (kid & 0x80000000u) cb = tbox[vals[kid & 0x7fffffffu]];
            else { cb = nbox[kid]; const uint32_t cd = ndepth[kid]; dep = cd > dep ? cd : dep; }   // (written before the sibling's release fence; read after this thread's acquire fence)
            if (c == 0) b = cb;
            else for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], cb.mn[a]); b.mx[a] = fmaxf(b.mx[a], cb.mx[a]); }
        }
        nbox[node] = b;
        ndepth[node] = real[node] ? dep + 1u : 0u;
        node = parent_of_inner[node];
    }
}

__device__ inline Box guarded(Box x) {                            // host/bvh_build.cpp refit(): the same arithmetic
    for (int a = 0; a < 3; ++a) {
        const float g = 1e-6f + 4e-7f * fmaxf(fabsf(x.mn[a]), fabsf(x.mx[a]));
        x.mn[a] -= g; x.mx[a] += g;
    }
    return x;
}
__global__ __launch_bounds__(256) void k_emit(const Box* __restrict__ tbox, const uint32_t* __restrict__ vals, int n_inner, const uint2* __restrict__ range,
                                              const uint2* __restrict__ kids, const uint32_t* __restrict__ real, const uint32_t* __restrict__ dense,
                                              const Box* __restrict__ nbox, uint32_t max_leaf, hrt_bvh_node* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !real[i]) return;
    const uint2 k = kids[i];
    hrt_bvh_node nd;
    memset(&nd, 0, sizeof(nd));
    for (int c = 0; c < 2; ++c) {
        const uint32_t kid = c == 0 ? k.x : k.y;
        Box cb; int32_t ref;
        if (kid & 0x80000000u) {
            const uint32_t p = kid & 0x7fffffffu;
            cb = tbox[vals[p]]; ref = (int32_t)~((p << 3) | 0u);
        } else {
            cb = nbox[kid];
            if (real[kid]) ref = (int32_t)dense[kid];
            else { const uint2 r = range[kid]; ref = (int32_t)~((r.x << 3) | (r.y - r.x)); }
        }
        cb = guarded(cb);
        if (c == 0) { nd.c0_min_x = cb.mn[0]; nd.c0_max_x = cb.mx[0]; nd.c0_min_y = cb.mn[1]; nd.c0_max_y = cb.mx[1]; nd.c0_min_z = cb.mn[2]; nd.c0_max_z = cb.mx[2]; nd.child0 = ref; }
        else { nd.c1_min_x = cb.mn[0]; nd.c1_max_x = cb.mx[0]; nd.c1_min_y = cb.mn[1]; nd.c1_max_y = cb.mx[1]; nd.c1_min_z = cb.mn[2]; nd.c1_max_z = cb.mx[2]; nd.child1 = ref; }
    }
    out[dense[i]] = nd;
}

struct DevBufs {
    void* p[32]; int n = 0;
    ~DevBufs() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
    template <class T> hipError_t get(T** out, size_t count) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e == hipSuccess) { p[n++] = q; *out = (T*)q; }
        return e;
    }
};

hrt_status lfail(hrt_status st, const std::string& msg) { hrt_set_last_error(msg.c_str()); return st; }

}  // namespace

#define LCHK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return lfail(e_ == hipErrorOutOfMemory ? HRT_ERR_OOM : HRT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {
struct DeviceGuard {      // the caller's current device is put back on every way out
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define LLAUNCH(name) do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return lfail(HRT_ERR_HIP, std::string(name " launch: ") + hipGetErrorString(e_)); } while (0)

hrt_status bvh_build_device_impl(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                                 uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out) {
    if (!tri_pos || !nodes_out || !n_nodes_out || !order_out || !depth_out) return lfail(HRT_ERR_INVALID, "hrt_bvh_build_device: NULL argument");
    if (max_leaf < 1 || max_leaf > 8) return lfail(HRT_ERR_INVALID, "hrt_bvh_build_device: max_leaf must be 1..8");
    if (n_tris <= max_leaf || n_tris >= (1u << 28)) return lfail(HRT_ERR_INVALID, "hrt_bvh_build_device: needs max_leaf < n_tris < 2^28");
    for (uint64_t k = 0; k < 9ull * n_tris; ++k)
        if (!(tri_pos[k] - tri_pos[k] == 0.0f)) return lfail(HRT_ERR_INVALID, "hrt_bvh_build_device: non-finite vertex");
    int n_dev = 0;
    LCHK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return lfail(HRT_ERR_NO_DEVICE, "hrt_bvh_build_device: no such device");
    DeviceGuard guard;
    LCHK(hipSetDevice(device));
    const int n = (int)n_tris, n_inner = n - 1;
    DevBufs bufs;
    float* d_pos; Box *d_tbox, *d_nbox; uint32_t *d_bounds, *d_vals_in, *d_vals, *d_pin, *d_pleaf, *d_real, *d_dense, *d_depth, *d_visits;
    uint64_t *d_keys_in, *d_keys; uint2 *d_range, *d_kids; hrt_bvh_node* d_out;
    LCHK(bufs.get(&d_pos, 9ull * n)); LCHK(bufs.get(&d_tbox, (size_t)n)); LCHK(bufs.get(&d_nbox, (size_t)n_inner)); LCHK(bufs.get(&d_bounds, 6));
    LCHK(bufs.get(&d_keys_in, (size_t)n)); LCHK(bufs.get(&d_keys, (size_t)n)); LCHK(bufs.get(&d_vals_in, (size_t)n)); LCHK(bufs.get(&d_vals, (size_t)n));
    LCHK(bufs.get(&d_range, (size_t)n_inner)); LCHK(bufs.get(&d_kids, (size_t)n_inner)); LCHK(bufs.get(&d_pin, (size_t)n_inner)); LCHK(bufs.get(&d_pleaf, (size_t)n));
    // real | dense | depth | visits in one allocation each would do; kept apart for clarity
    LCHK(bufs.get(&d_real, (size_t)n_inner)); LCHK(bufs.get(&d_dense, (size_t)n_inner)); LCHK(bufs.get(&d_depth, (size_t)n_inner)); LCHK(bufs.get(&d_visits, (size_t)n_inner));
    struct OwnStream {      // a stream of its own, not the legacy NULL stream (which serialises with every other stream of the process)
        hipStream_t s = nullptr;
        ~OwnStream() { if (s) (void)hipStreamDestroy(s); }
    } own;
    LCHK(hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking));
    hipStream_t stream = own.s;
    LCHK(hipMemcpyAsync(d_pos, tri_pos, 9ull * n * sizeof(float), hipMemcpyHostToDevice, stream));
    const uint32_t init_bounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    LCHK(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
    LCHK(hipMemsetAsync(d_visits, 0, (size_t)n_inner * sizeof(uint32_t), stream));
    const unsigned gb = (unsigned)((n + 255) / 256), gi = (unsigned)((n_inner + 255) / 256);
    hipLaunchKernelGGL(k_prep, dim3(gb), dim3(256), 0, stream, d_pos, (uint32_t)n, d_tbox, d_bounds);
    LLAUNCH("k_prep");
    hipLaunchKernelGGL(k_codes, dim3(gb), dim3(256), 0, stream, d_tbox, (uint32_t)n, d_bounds, d_keys_in, d_vals_in);
    LLAUNCH("k_codes");
    {
        size_t tmp_bytes = 0;
        LCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals, (size_t)n, 0, 63, stream));
        char* tmp; LCHK(bufs.get(&tmp, tmp_bytes));
        LCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals, (size_t)n, 0, 63, stream));
    }
    hipLaunchKernelGGL(k_hierarchy, dim3(gi), dim3(256), 0, stream, d_keys, n, d_range, d_kids, d_pin, d_pleaf);
    LLAUNCH("k_hierarchy");
    hipLaunchKernelGGL(k_flag, dim3(gi), dim3(256), 0, stream, d_range, n_inner, max_leaf, d_real);
    LLAUNCH("k_flag");
    {
        size_t tmp_bytes = 0;
        LCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, d_real, d_dense, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
        char* tmp; LCHK(bufs.get(&tmp, tmp_bytes));
        LCHK(rocprim::exclusive_scan(tmp, tmp_bytes, d_real, d_dense, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), stream));
    }
    hipLaunchKernelGGL(k_refit, dim3(gb), dim3(256), 0, stream, d_tbox, d_vals, n, d_kids, d_pin, d_pleaf, d_real, d_nbox, d_depth, d_visits);
    LLAUNCH("k_refit");
    LCHK(bufs.get(&d_out, (size_t)n_inner));
    hipLaunchKernelGGL(k_emit, dim3(gi), dim3(256), 0, stream, d_tbox, d_vals, n_inner, d_range, d_kids, d_real, d_dense, d_nbox, max_leaf, d_out);
    LLAUNCH("k_emit");
    uint32_t last_real = 0, last_dense = 0, root_depth = 0;
    LCHK(hipMemcpyAsync(&last_real, d_real + (n_inner - 1), 4, hipMemcpyDeviceToHost, stream));
    LCHK(hipMemcpyAsync(&last_dense, d_dense + (n_inner - 1), 4, hipMemcpyDeviceToHost, stream));
    LCHK(hipMemcpyAsync(&root_depth, d_depth, 4, hipMemcpyDeviceToHost, stream));
    LCHK(hipStreamSynchronize(stream));
    const uint32_t n_nodes = last_dense + last_real;
    LCHK(hipMemcpy(nodes_out, d_out, (size_t)n_nodes * sizeof(hrt_bvh_node), hipMemcpyDeviceToHost));
    LCHK(hipMemcpy(order_out, d_vals, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *n_nodes_out = n_nodes;
    *depth_out = (int32_t)root_depth;
    return HRT_OK;
}
}  // namespace

// include/hrt.h: "never throws" -- std::string, rocprim and the buffer holder can
extern "C" hrt_status hrt_bvh_build_device(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                                           uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out) {
    try {
        return bvh_build_device_impl(device, tri_pos, n_tris, max_leaf, nodes_out, n_nodes_out, order_out, depth_out);
    } catch (const std::bad_alloc&) { return lfail(HRT_ERR_OOM, "hrt_bvh_build_device: out of host memory"); }
    catch (const std::exception& e) { return lfail(HRT_ERR_INVALID, std::string("hrt_bvh_build_device: ") + e.what()); }
    catch (...) { return lfail(HRT_ERR_INVALID, "hrt_bvh_build_device: unknown C++ exception"); }
}
