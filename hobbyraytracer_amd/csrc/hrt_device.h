// hrt_device.h — device-side restatement of the reference's hit / scatter /
// texture code over the FLATTENED scene (no virtual dispatch, no pointers to
// objects): one function per reference method, cited at each definition.
//
// Arithmetic rule: every value that reaches a hitRecord, a scattered ray or a
// pixel is computed with the same IEEE operations in the same order as the
// reference lines cited (and therefore as oracle/oracle.cpp), with
// -ffp-contract=off.  The only place that is free to differ is BVH *culling*
// (bvh_traverse): a culled box can only remove triangles the reference would
// also have rejected, because the final acceptance of a candidate uses the
// reference's own leaf-level box test (accept_box below).
#pragma once
#include "hrt_rng.h"
#include "../../include/hrt.h"

namespace hrt {

#ifndef HRT_STACK_DEPTH
#define HRT_STACK_DEPTH 32
#endif
#define HRT_BLOCK 256

struct DScene {
    const hrt_prim* prims;
    const hrt_material* mats;
    const hrt_texture* texs;
    const hrt_mesh* meshes;
    const uint4* qnodes;     // 2 x uint4 per node: 16-bit grid culling record (hrt_pack.h pack_nodes)
    const float4* grids;     // 2 x float4 per mesh: grid origin.xyz_, step.xyz_
    const float4* tri_pos;   // 3 x float4 per triangle: v0.xyz_, v1.xyz_, v2.xyz_
    const float4* tri_attr;  // 4 x float4 per triangle: n0.xyz uv0.x | n1.xyz uv0.y | n2.xyz uv1.x | uv1.y uv2.x uv2.y _
    const float4* tri_box;   // 2 x float4 per triangle: reference leaf-level box (min.xyz_, max.xyz_)
    // The reference's own tree, threaded in preorder (hrt_pack.h pack_ref_tree), for ref_walk below:
    const uint4* rnodes;     // 2 x uint4 per BVHNode: min.xyz skip | max.xyz (inner: 0x80000000 | right child; leaf: (first << 1) | (count - 1))
    const float4* rtris;     // 3 x float4 per triangle in the reference's test order: v0.xyz index | v1.xyz_ | v2.xyz_
    const uint4* rmesh;      // per mesh: first node, node count, first triangle in rtris, 0
    float q4_route_a2;       // rays with (|d| / |d[kZ]|)^2 above this take ref_walk (q4_risky)
    int32_t ref_fold_all;    // tests: bvh_traverse (megakernel, test kernels) takes ref_fold instead of ref_walk where it is valid
    int32_t stale_ff;        // some mesh stands in the world list without a wrapper and has a Dielectric material: its hits take
                             // hitRecord::frontFace from the previous successful object of the walk (WorldHit, "stale frontFace")
    const uint8_t* texels_u8;
    const float* texels_f32;
    int32_t n_prims;
    int32_t background_tex;
    // Copies of the four small tables for PER-LANE indexed lookups (hit prim -> mesh -> material ->
    // texture is a chain of dependent gathers): kernels stage them in LDS once per workgroup
    // (stage_tables) so each hop costs an LDS read (~64 cycles) instead of an L2 round trip.  Wave-uniform
    // walks over the tables (the world-list loop) keep using the global pointers above, which the compiler
    // turns into scalar loads.  Host code and tables too large for LDS point these at the global arrays.
    const hrt_prim* lprims;
    const hrt_material* lmats;
    const hrt_texture* ltexs;
    const hrt_mesh* lmeshes;
    int32_t n_mats, n_texs, n_meshes;
};

// Wave-uniform walks over the small scene tables (the world-list loop, the mesh prim's wrapper chain) must be
// SCALAR loads.  Through a plain global pointer the compiler may not scalarise them in kernels that also store
// (the scalar cache is not coherent with vector stores), and each field then costs a full vector-memory round
// trip -- measured: 41 % of k_wf_shade.  The tables are never written by a kernel, so view them through the
// constant address space, whose loads are invariant by definition (s_load_*).  Host builds: plain pointers.
#if defined(__HIPCC__)
#define HRT_CONST_AS __attribute__((address_space(4)))
template <class T> __device__ inline const HRT_CONST_AS T* uniform_table(const T* p) { return (const HRT_CONST_AS T*)p; }
#else
#define HRT_CONST_AS
template <class T> inline const T* uniform_table(const T* p) { return p; }
#endif

#define HRT_TABLE_LDS_BYTES 12288
#if defined(__HIPCC__)
// Copies prims / mats / texs / meshes into `buf` (`budget` bytes of LDS, 16-byte aligned) when they
// fit, and repoints sc.l*; every thread of the block must call it (it ends with a barrier).
__device__ inline void stage_tables(DScene& sc, uint32_t* buf, uint32_t budget = HRT_TABLE_LDS_BYTES) {
    const uint32_t wp = (uint32_t)(sc.n_prims * sizeof(hrt_prim) + 15) / 16 * 4, wm = (uint32_t)(sc.n_mats * sizeof(hrt_material) + 15) / 16 * 4;
    const uint32_t wt = (uint32_t)(sc.n_texs * sizeof(hrt_texture) + 15) / 16 * 4, wx = (uint32_t)(sc.n_meshes * sizeof(hrt_mesh) + 15) / 16 * 4;
    if ((wp + wm + wt + wx) * 4u <= budget) {
        const uint32_t* src[4] = {(const uint32_t*)sc.prims, (const uint32_t*)sc.mats, (const uint32_t*)sc.texs, (const uint32_t*)sc.meshes};
        const uint32_t words[4] = {(uint32_t)(sc.n_prims * sizeof(hrt_prim)) / 4, (uint32_t)(sc.n_mats * sizeof(hrt_material)) / 4,
                                   (uint32_t)(sc.n_texs * sizeof(hrt_texture)) / 4, (uint32_t)(sc.n_meshes * sizeof(hrt_mesh)) / 4};
        const uint32_t off[4] = {0, wp, wp + wm, wp + wm + wt};
        for (int k = 0; k < 4; ++k)
            for (uint32_t i = threadIdx.x; i < words[k]; i += blockDim.x) buf[off[k] + i] = src[k][i];
        sc.lprims = (const hrt_prim*)(buf + off[0]);
        sc.lmats = (const hrt_material*)(buf + off[1]);
        sc.ltexs = (const hrt_texture*)(buf + off[2]);
        sc.lmeshes = (const hrt_mesh*)(buf + off[3]);
    }
    __syncthreads();
}
#endif

struct DRec {  // hitRecord (hittable.h:8-25)
    vec3 p, normal;
    float t, u, v;
    int32_t mat;
    bool frontFace;
};

struct DCounters {
    uint32_t box_tests, tri_tests;
};

// ---- WorldHit::sub of a MESH hit = mesh-local triangle index in the low 28 bits (hrt_scene_create refuses meshes of 2^28 or more
// triangles) + the flags HRT_SUB_TIE_UNSETTLED / HRT_SUB_WRAPPERLESS / HRT_SUB_STALE_BACK above it (defined where they are
// produced, below).  EVERY reader of a triangle index goes through sub_tri: an index that still carries a flag addresses
// tri_pos / tri_attr 2^28..2^30 records past the mesh (a round-2 ablation build that skipped the settling of
// HRT_SUB_TIE_UNSETTLED died of exactly that: DESIGN.md 6.1, "the NO_SETTLE fault").
#define HRT_SUB_TRI_MASK 0x0fffffff
__device__ inline int sub_tri(int sub) { return sub & HRT_SUB_TRI_MASK; }
__device__ inline int sub_flags(int sub) { return sub & ~HRT_SUB_TRI_MASK; }

// -DHRT_DEBUG_BOUNDS (tests/tools/debug_bounds.sh): every table index a hit record is built from is checked against its table;
// a violation is COUNTED (hrt_debug_bounds_violations) and the index replaced by 0, so that the run goes on and ends with a
// number instead of a GPU memory fault.
#if defined(HRT_DEBUG_BOUNDS) && defined(__HIPCC__)
#define HRT_BOUNDS_SLOTS 8      // 0 triangle of a mesh, 1 prim, 2 material, 3 texture, 4 mesh, 5 frontFace source (k_wf_stale), 6 reference-tree node
static __device__ unsigned long long g_hrt_bounds_violations[HRT_BOUNDS_SLOTS];
#if defined(__HIP_DEVICE_COMPILE__)
#define HRT_BOUNDS(slot, idx, n) do { if ((unsigned long long)(long long)(idx) >= (unsigned long long)(n)) { atomicAdd(&g_hrt_bounds_violations[slot], 1ull); (idx) = 0; } } while (0)
#else
#define HRT_BOUNDS(slot, idx, n) do { } while (0)
#endif
#else
#define HRT_BOUNDS(slot, idx, n) do { } while (0)
#endif

__device__ inline void set_face_normal(DRec& rec, vec3 rdir, vec3 outward) {  // hittable.h:21-24
    rec.frontFace = dot(rdir, outward) < 0;
    rec.normal = rec.frontFace ? outward : -outward;
}

// ------------------------------------------------------------------ wrappers (ray side)
// translate.cpp:9, scale.cpp:13-16, rotateQuat.cpp:47-52, rotateY.cpp:46-54
template <class XF>
__device__ inline void xf_apply(const XF& x, vec3& o, vec3& d, uint32_t quirks) {
    if (x.kind == HRT_XF_TRANSLATE) {
        o = o - vec3(x.v[0], x.v[1], x.v[2]);
    } else if (x.kind == HRT_XF_SCALE) {
        vec3 f(x.v[0], x.v[1], x.v[2]);
        o = o / f; d = d / f;
    } else if (x.kind == HRT_XF_ROTATE_QUAT) {
        quat q; q.x = x.v[0]; q.y = x.v[1]; q.z = x.v[2]; q.w = x.v[3];
        quat inv = conjugate(q);
        o = rotate(inv, o);
        d = rotate(inv, d);
        if (quirks & HRT_Q1_ROTQ_NORMALIZE) d = normalize(d);
    } else {
        float s = x.v[0], c = x.v[1];
        vec3 o2 = o, d2 = d;
        o2.x = c * o.x - s * o.z; o2.z = s * o.x + c * o.z;
        d2.x = c * d.x - s * d.z; d2.z = s * d.x + c * d.z;
        o = o2; d = d2;
    }
}
// record side: translate.cpp:15-16, scale.cpp:23-24, rotateQuat.cpp:60-63, rotateY.cpp:61-73
// `ldir` = direction of the ray this wrapper handed to its child.
__device__ inline void xf_unapply(const hrt_xform& x, DRec& rec, vec3 ldir) {
    if (x.kind == HRT_XF_TRANSLATE) {
        rec.p = rec.p + vec3(x.v[0], x.v[1], x.v[2]);
        set_face_normal(rec, ldir, rec.normal);
    } else if (x.kind == HRT_XF_SCALE) {
        rec.p = rec.p * vec3(x.v[0], x.v[1], x.v[2]);
        set_face_normal(rec, ldir, rec.normal);
    } else if (x.kind == HRT_XF_ROTATE_QUAT) {
        quat q; q.x = x.v[0]; q.y = x.v[1]; q.z = x.v[2]; q.w = x.v[3];
        rec.p = rotate(q, rec.p);
        rec.normal = rotate(q, rec.normal);
        set_face_normal(rec, ldir, rec.normal);
    } else {
        float s = x.v[0], c = x.v[1];
        vec3 p = rec.p, n = rec.normal;
        p.x = c * rec.p.x + s * rec.p.z; p.z = -s * rec.p.x + c * rec.p.z;
        n.x = c * rec.normal.x + s * rec.normal.z; n.z = -s * rec.normal.x + c * rec.normal.z;
        rec.p = p;
        set_face_normal(rec, ldir, n);
    }
}

// ------------------------------------------------------------------ analytic primitives
// aarect.h:12-39 / 59-86 / 106-133.  `axis` = the constant axis (0:YZ, 1:XZ, 2:XY).
// p = a0,a1,b0,b1,k in the reference's member order.
template <class P>
__device__ inline bool rect_hit(int axis, P p, vec3 o, vec3 d, float t_min, float t_max, float& t_out) {
    float ok, dk, oa, da, ob, db;
    if (axis == 0) { ok = o.x; dk = d.x; oa = o.y; da = d.y; ob = o.z; db = d.z; }
    else if (axis == 1) { ok = o.y; dk = d.y; oa = o.x; da = d.x; ob = o.z; db = d.z; }
    else { ok = o.z; dk = d.z; oa = o.x; da = d.x; ob = o.y; db = d.y; }
    float t = (p[4] - ok) / dk;
    if (t < t_min || t > t_max) return false;
    float a = oa + t * da;
    float b = ob + t * db;
    if (a < p[0] || a > p[1] || b < p[2] || b > p[3]) return false;
    t_out = t;
    return true;
}
__device__ inline void rect_rec(int axis, const float* p, vec3 o, vec3 d, float t, DRec& rec) {
    float oa, da, ob, db;
    if (axis == 0) { oa = o.y; da = d.y; ob = o.z; db = d.z; }
    else if (axis == 1) { oa = o.x; da = d.x; ob = o.z; db = d.z; }
    else { oa = o.x; da = d.x; ob = o.y; db = d.y; }
    float a = oa + t * da;
    float b = ob + t * db;
    rec.u = (a - p[0]) / (p[1] - p[0]);
    rec.v = (b - p[2]) / (p[3] - p[2]);
    rec.t = t;
    vec3 n = axis == 0 ? vec3(1, 0, 0) : (axis == 1 ? vec3(0, 1, 0) : vec3(0, 0, 1));
    set_face_normal(rec, d, n);
    rec.p = o + (t * d);
}
__device__ inline int rect_axis(int kind) { return kind == HRT_PRIM_YZ_RECT ? 0 : (kind == HRT_PRIM_XZ_RECT ? 1 : 2); }

// sphere.cpp:20-36
template <class P>
__device__ inline bool sphere_hit(P p, vec3 o, vec3 d, float t_min, float t_max, float& t_out) {
    vec3 center(p[0], p[1], p[2]);
    float radius = p[3];
    vec3 oc = o - center;
    float a = length(d) * length(d);
    float half_b = dot(oc, d);
    float c = length(oc) * length(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return false;
    }
    t_out = root;
    return true;
}
// sphere.cpp:38-46 + getSphereUV (sphere.cpp:4-18)
__device__ inline void sphere_rec(const float* p, vec3 o, vec3 d, float t, DRec& rec) {
    const float pi = 3.14159265358979323846264338327950288f;
    vec3 center(p[0], p[1], p[2]);
    rec.t = t;
    rec.p = o + (t * d);
    vec3 outward = (rec.p - center) / p[3];
    set_face_normal(rec, d, outward);
    float theta = gacos(-outward.y);
    float phi = gatan2(-outward.z, outward.x) + pi;
    rec.u = phi / (2 * pi);
    rec.v = theta / pi;
}

// box.h:27-55: six rects in constructBox order, HittableList::hit semantics.
template <class P>
__device__ inline void box_side(P p, int side, int& axis, float* rp) {
    // p = min.xyz, max.xyz
    if (side < 2) { axis = 2; rp[0] = p[0]; rp[1] = p[3]; rp[2] = p[1]; rp[3] = p[4]; rp[4] = side == 0 ? p[5] : p[2]; }
    else if (side < 4) { axis = 1; rp[0] = p[0]; rp[1] = p[3]; rp[2] = p[2]; rp[3] = p[5]; rp[4] = side == 2 ? p[4] : p[1]; }
    else { axis = 0; rp[0] = p[1]; rp[1] = p[4]; rp[2] = p[2]; rp[3] = p[5]; rp[4] = side == 4 ? p[3] : p[0]; }
}
template <class P>
__device__ inline bool box_hit(P p, vec3 o, vec3 d, float t_min, float t_max, float& t_out, int& side_out) {
    bool any = false;
    float closest = t_max;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        int axis; float rp[5];
        box_side(p, s, axis, rp);
        float t;
        if (rect_hit(axis, rp, o, d, t_min, closest, t)) { any = true; closest = t; side_out = s; }
    }
    t_out = closest;
    return any;
}
__device__ inline void box_rec(const float* p, vec3 o, vec3 d, float t, int side, DRec& rec) {
    int axis; float rp[5];
    box_side(p, side, axis, rp);
    rect_rec(axis, rp, o, d, t, rec);
}

// triangle.cpp:4-40 -- Triangle::hit exactly as written: pV, v0v1 and tV are NORMALISED before use, so u, v and t are not
// the barycentrics and the distance of Moeller-Trumbore (they only coincide for special rays); back faces are culled
// (d < 0.0001).  rec.u / rec.v are those values; the normal is the un-normalised face normal, re-faced (hittable.h:19-24).
template <class P>
__device__ inline bool triangle_eval(P p, vec3 o, vec3 d, float& t, float& u, float& v) {
    const vec3 v0(p[0], p[1], p[2]), v1(p[3], p[4], p[5]), v2(p[6], p[7], p[8]);
    const vec3 v0v1 = v1 - v0;
    const vec3 v0v2 = v2 - v0;
    const vec3 pV = normalize(cross(d, v0v2));
    const float dd = dot(normalize(v0v1), pV);
    if (dd < 0.0001f) return false;
    if (fabsf(dd) < 0.0001f) return false;
    const float invD = 1.0f / dd;
    const vec3 tV = normalize(o - v0);
    u = dot(tV, pV) * invD;
    if (u < 0 || u > 1) return false;
    const vec3 qV = cross(tV, normalize(v0v1));
    v = dot(normalize(d), qV) * invD;
    if (v < 0 || u + v > 1) return false;
    t = dot(v0v2, qV) * invD;
    return true;
}
template <class P>
__device__ inline bool triangle_hit(P p, vec3 o, vec3 d, float t_min, float t_max, float& t_out) {
    float t, u, v;
    if (!triangle_eval(p, o, d, t, u, v)) return false;
    if (t < t_min) return false;
    if (t > t_max) return false;
    t_out = t;
    return true;
}
template <class P>
__device__ inline void triangle_rec(P p, vec3 o, vec3 d, DRec& rec) {
    float t = 0.0f, u = 0.0f, v = 0.0f;
    triangle_eval(p, o, d, t, u, v);
    rec.t = t; rec.u = u; rec.v = v;
    rec.p = o + (t * d);
    const vec3 v0(p[0], p[1], p[2]), v1(p[3], p[4], p[5]), v2(p[6], p[7], p[8]);
    set_face_normal(rec, d, cross(v1 - v0, v2 - v0));
}

template <class P>
__device__ inline bool boundary_hit(int kind, P p, vec3 o, vec3 d, float t_min, float t_max, float& t) {
    int side;
    if (kind == HRT_PRIM_SPHERE) return sphere_hit(p, o, d, t_min, t_max, t);
    return box_hit(p, o, d, t_min, t_max, t, side);
}
// constantMedium.cpp:4-38
template <class PR>
__device__ inline bool medium_hit(const PR& pr, uint32_t prim_index, vec3 o, vec3 d, float t_min, float t_max,
                                  const rng_ctx& ctx, float& t_out) {
    const float INF = __builtin_huge_valf();
    float t1, t2;
    if (!boundary_hit(pr.boundary_kind, pr.p, o, d, -INF, INF, t1)) return false;
    if (!boundary_hit(pr.boundary_kind, pr.p, o, d, t1 + 0.0001f, INF, t2)) return false;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0) t1 = 0;
    const float ray_length = length(d);
    const float distance_inside_boundary = (t2 - t1) * ray_length;
    u32x4 u = rng_draw(ctx, RNG_MEDIUM, prim_index);
    const float negInvDensity = -1 / pr.density;
    const float hit_distance = negInvDensity * glog(linear_rand(u.x, 0.0f, 1.0f));
    if (hit_distance > distance_inside_boundary) return false;
    t_out = t1 + hit_distance / ray_length;
    return true;
}

// ------------------------------------------------------------------ ITriangle::hit (triangle.cpp:57-131)
struct TriRay {          // per-ray constants of the triangle test
    vec3 o;
    float sX, sY, sZ;
    int kZ;
};
__device__ inline TriRay tri_ray_setup(vec3 o, vec3 d, uint32_t quirks) {
    TriRay tr;
    tr.o = o;
    int kZ;
    if (quirks & HRT_Q4_SHEAR_FROM_ORIGIN) {
        kZ = o.x > o.z ? (o.x > o.y ? 0 : 1) : 2;  // triangle.cpp:70
    } else {
        float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
        kZ = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    }
    tr.kZ = kZ;
    // d = { d[kX], d[kY], d[kZ] }
    float dx = kZ == 2 ? d.x : (kZ == 0 ? d.y : d.z);
    float dy = kZ == 2 ? d.y : (kZ == 0 ? d.z : d.x);
    float dz = kZ == 2 ? d.z : (kZ == 0 ? d.x : d.y);
    tr.sX = -dx / dz;
    tr.sY = -dy / dz;
    tr.sZ = 1.0f / dz;
    return tr;
}
__device__ inline vec3 tri_permute(vec3 v, int kZ) {
    return vec3(kZ == 2 ? v.x : (kZ == 0 ? v.y : v.z), kZ == 2 ? v.y : (kZ == 0 ? v.z : v.x),
                kZ == 2 ? v.z : (kZ == 0 ? v.x : v.y));
}
struct TriEval { float e0, e1, e2, det, tScaled; };
// triangle.cpp:64-105.  Returns false on the edge / determinant rejections.
__device__ inline bool tri_eval(const TriRay& tr, vec3 v0, vec3 v1, vec3 v2, TriEval& ev) {
    vec3 p0t = tri_permute(v0 - tr.o, tr.kZ);
    vec3 p1t = tri_permute(v1 - tr.o, tr.kZ);
    vec3 p2t = tri_permute(v2 - tr.o, tr.kZ);
    p0t.x = p0t.x + tr.sX * p0t.z; p0t.y = p0t.y + tr.sY * p0t.z;
    p1t.x = p1t.x + tr.sX * p1t.z; p1t.y = p1t.y + tr.sY * p1t.z;
    p2t.x = p2t.x + tr.sX * p2t.z; p2t.y = p2t.y + tr.sY * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z = p0t.z * tr.sZ; p1t.z = p1t.z * tr.sZ; p2t.z = p2t.z * tr.sZ;
    ev.e0 = e0; ev.e1 = e1; ev.e2 = e2; ev.det = det;
    ev.tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    return true;
}
// The leaf-level BVHNode::hit box test of the reference tree (aabb.h:26-39 on
// the box of bvh.cpp:52-60), evaluated only for candidates that passed the
// triangle test.  std::min(a, b) is (b < a) ? b : a and std::max(a, b) is (a < b) ? b : a -- the ARGUMENT ORDER of
// aabb.h:29-34 is kept, because it decides what a NaN does: a NaN t_max (the hit so far came from a degenerate primitive)
// is replaced by the slab's own exit, a NaN slab (NaN ray) replaces t_min and t_max and the box passes.
__device__ inline bool accept_box(float4 bmn, float4 bmx, vec3 o, vec3 d, float t_min, float t_max) {
    {
        float ta = (bmn.x - o.x) / d.x, tb = (bmx.x - o.x) / d.x;
        float t0 = (tb < ta) ? tb : ta; float t1 = (ta < tb) ? tb : ta;
        t_min = (t0 < t_min) ? t_min : t0; t_max = (t_max < t1) ? t_max : t1;   // std::max(t0, t_min), std::min(t1, t_max)
        if (t_max <= t_min) return false;
    }
    {
        float ta = (bmn.y - o.y) / d.y, tb = (bmx.y - o.y) / d.y;
        float t0 = (tb < ta) ? tb : ta; float t1 = (ta < tb) ? tb : ta;
        t_min = (t0 < t_min) ? t_min : t0; t_max = (t_max < t1) ? t_max : t1;   // std::max(t0, t_min), std::min(t1, t_max)
        if (t_max <= t_min) return false;
    }
    {
        float ta = (bmn.z - o.z) / d.z, tb = (bmx.z - o.z) / d.z;
        float t0 = (tb < ta) ? tb : ta; float t1 = (ta < tb) ? tb : ta;
        t_min = (t0 < t_min) ? t_min : t0; t_max = (t_max < t1) ? t_max : t1;   // std::max(t0, t_min), std::min(t1, t_max)
        if (t_max <= t_min) return false;
    }
    return true;
}

// ------------------------------------------------------------------ BVHNode::hit verbatim (bvh.cpp:69-78) for the rays of quirk Q-4
// ITriangle::hit takes its shear axis from the ray ORIGIN (triangle.cpp:70, Q-4).  When the direction component on that
// axis is small against |d|, the sheared coordinates are stretched by A = |d| / |d[kZ]| and the computed t of a triangle
// seen edge-on loses its digits: at A = 200 a silhouette triangle truly met at t = 1.34937 reports 1.34900 -- in FRONT of
// its own leaf box, and of the neighbour (1.34901) that is truly closer.  The reference keeps whichever of the two ITS
// walk meets first: a first hit shrinks t_max, and BVHNode::hit (bvh.cpp:71) then drops every box that starts beyond it,
// although the triangle inside would have reported a still smaller t.  No other tree can reproduce that (measured on the
// teapot against the oracle, tests/tools/q4_study.py: one differing hit in 1.4e6 for A in [180, 320), one in 10^4 at
// A = 2000, one in 40 beyond 3e4; none in 5.6e6 hits below A = 178), so such rays walk the reference's own tree, node by
// node in its order, with its box test and its triangle test as written.  The tree is threaded (pack_ref_tree): the walk
// is a loop over node indices without a stack -- a node whose box fails, or a leaf, continues at its `skip`.
// Default threshold A = 128: 0.8 % of isotropic rays.
#define HRT_Q4_ROUTE_A_DEFAULT 128.0f
__device__ inline bool q4_risky(const TriRay& tr, vec3 d, uint32_t quirks, float route_a2) {
    // (sZ = 1 / d[kZ]; an exactly zero component gives inf > a2: routed, and every triangle test then yields NaN as in the
    //  reference; a NaN direction compares false and stays with the culling traversal and its NaN rules.)
    return (quirks & HRT_Q4_SHEAR_FROM_ORIGIN) && tr.sZ * tr.sZ * dot(d, d) > route_a2;
}
// Returns the winning triangle (mesh-local index) or -1; t_out = its t.  t_max may be +inf, never NaN (mesh_t_max).
// (The box / triangle tests of these rays are NOT added to the STATS counters: the three implementations -- this walk, ref_fold,
//  the pipeline's level walk -- visit different numbers of nodes, and the counters are defined by the culling traversal.)
template <bool STATS>
__device__ inline int ref_walk(const uint4* __restrict__ rn, const float4* __restrict__ rt, uint32_t node_count, vec3 o, vec3 d,
                               const TriRay& tr, float t_min, float t_max, uint32_t quirks, float& t_out, DCounters& cnt) {
    int best = -1;
    uint32_t i = 0;
    while (i < node_count) {
        const uint4 A = rn[2 * i], B = rn[2 * i + 1];
        float4 bmn, bmx;
        bmn.x = u2f(A.x); bmn.y = u2f(A.y); bmn.z = u2f(A.z); bmn.w = 0.0f;
        bmx.x = u2f(B.x); bmx.y = u2f(B.y); bmx.z = u2f(B.z); bmx.w = 0.0f;
        if (!accept_box(bmn, bmx, o, d, t_min, t_max)) { i = A.w; continue; }      // bvh.cpp:71
        if (B.w & 0x80000000u) { ++i; continue; }                                    // bvh.cpp:74: left is the next node
        const uint32_t first = B.w >> 1, count = (B.w & 1u) + 1u;
        for (uint32_t k = 0; k < count; ++k) {                                      // bvh.cpp:74-75 over ITriangles
            const float4 q0 = rt[3 * (first + k) + 0], q1 = rt[3 * (first + k) + 1], q2 = rt[3 * (first + k) + 2];
            TriEval ev;
            if (!tri_eval(tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev)) continue;
            const float lim = t_max * ev.det;                                        // triangle.cpp:106-109
            if (ev.det < 0 && (ev.tScaled >= 0 || ev.tScaled < lim)) continue;
            else if (ev.det > 0 && (ev.tScaled <= 0 || ev.tScaled > lim)) continue;
            const float invDet = 1 / ev.det;
            const float t = ev.tScaled * invDet;
            if (!(quirks & HRT_Q2_TRI_NO_TMIN) && t < t_min) continue;
            t_max = t;                                                               // bvh.cpp:75: right gets rec.t
            best = (int)f2u(q0.w);
        }
        i = A.w;
    }
    t_out = t_max;
    return best;
}

// What ref_walk computes, without the tree: a fold over ALL triangles in the tree's depth-first order (rt), each candidate
// gated by the box of its lowest node only (tbox: mesh-local triangle -> that box + its (node << 1) | side code).  The
// wavefront pipeline runs this fold with a whole wave per ray (hrt_hip.hip wf_ref_fold, where the argument is spelled
// out); this is the same thing one triangle after the other, for the CPU check of the two against each other
// (tests/test_flat_vs_oracle_cpu.py) and for the test kernels (DScene::ref_fold_all).  Not valid for rays with an exactly
// zero direction component.
template <bool STATS>
__device__ inline int ref_fold(const float4* __restrict__ rt, const float4* __restrict__ tbox, uint32_t tri_count, vec3 o, vec3 d, const TriRay& tr,
                               float t_min, float t_max, uint32_t quirks, float& t_out, DCounters& cnt) {
    int best = -1;
    uint32_t cur_node = 0xffffffffu;
    bool node_ok = false;
    for (uint32_t p = 0; p < tri_count; ++p) {
        const float4 q0 = rt[3 * p + 0], q1 = rt[3 * p + 1], q2 = rt[3 * p + 2];
        TriEval ev;
        if (!tri_eval(tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev)) continue;
        if ((ev.det < 0 && ev.tScaled >= 0) || (ev.det > 0 && ev.tScaled <= 0)) continue;
        const uint32_t ti = f2u(q0.w);
        const float4 bmn = tbox[2 * ti], bmx = tbox[2 * ti + 1];
        const uint32_t node = f2u(bmn.w) >> 1;
        if (node != cur_node) {
            cur_node = node;
            node_ok = accept_box(bmn, bmx, o, d, t_min, t_max);
        }
        if (!node_ok) continue;
        const float lim = t_max * ev.det;
        if (ev.det < 0 && ev.tScaled < lim) continue;
        if (ev.det > 0 && ev.tScaled > lim) continue;
        const float t = ev.tScaled * (1 / ev.det);
        if (!(quirks & HRT_Q2_TRI_NO_TMIN) && t < t_min) continue;
        t_max = t;
        best = (int)ti;
    }
    t_out = t_max;
    return best;
}

// ------------------------------------------------------------------ Mesh::hit -> BVHNode::hit (mesh.cpp:43-46, bvh.cpp:69-78)
// Flattened 2-wide BVH walked by a RESUMABLE state machine: trav_step() advances one lane by one node or
// one leaf, so the megakernel (bvh_traverse below) and the wavefront traversal kernel (which refills
// finished lanes with new rays between steps) run the very same code.  Per-lane stack in LDS
// (stack[depth][thread]: lanes of a wave hit distinct banks whatever their depths).
struct MeshRay {          // a ray in mesh space plus its per-ray constants
    vec3 o, d;
    TriRay tr;            // exact-arithmetic constants of the triangle test
    float idx, idy, idz;  // culling-only constants (free to differ from the reference: see header): 1/d,
    float gx, gy, gz;     // and the slab test in the mesh's node-grid coordinates: t = q * g - o_ (q = 16-bit grid index);
                          // bit 4 of each g says "negative": the rotation that brings the NEAR face of a node word down (node_test)
    float reach;          // mesh_ray_reach (only until mesh_ray_grid has run)
    float oxn, oxf, oyn, oyf, ozn, ozf;   // (o - origin) / d per axis, shifted for the NEAR / FAR face of a box (see mesh_ray_grid)
};
// Q-4: the shear axis comes from the ray ORIGIN (triangle.cpp:70), so the t of ITriangle::hit is only good to about
// kappa * eps * M / |d[kZ]| (M = the size of the coordinates that cancel in v - o; kappa in the hundreds for a thin triangle
// met near an edge).  A triangle of the surface the ray has just LEFT -- truly behind the origin by less than that --
// then reports a small positive t, and the reference takes it if the box of its lowest node (the union with its
// neighbour in the tree) reaches the ray: at |d| / |d[kZ]| = 45 a triangle 3.9e-4 behind the origin came out at
// t = +2.0e-5 (tests/tools/gpu_fuzz_meshes.py; one path in 2e8).  The culling boxes here are this tree's own, and the ray
// starts outside them: so every slab is widened by that reach (in t, both ways: the same shift as above), which lets the
// walk look that far behind the origin and beyond a box's far side.  1e-4 * M / |d[kZ]|: 20x what was seen; a few 1e-4
// units for an ordinary ray, 1e-2 at the |d| / |d[kZ]| = 128 beyond which rays walk the reference's tree anyway.
__device__ inline float mesh_ray_reach(vec3 o, float sZ, float4 origin, float4 step, uint32_t quirks) {
    if (!(quirks & HRT_Q4_SHEAR_FROM_ORIGIN)) return 0.0f;
    const float M = fmaxf(fmaxf(fmaxf(fabsf(o.x), fabsf(origin.x) + 65535.0f * step.x), fmaxf(fabsf(o.y), fabsf(origin.y) + 65535.0f * step.y)),
                          fmaxf(fabsf(o.z), fabsf(origin.z) + 65535.0f * step.z));
    const float reach = 1e-4f * fabsf(sZ) * M;
    return reach < 1e30f ? reach : 0.0f;           // (NaN rays keep their own rules; d[kZ] == 0 walks the reference's tree)
}
// box = origin + q * step  =>  t = (box - o) / d = q * (step / d) - (o - origin) / d
// `reach`: every slab is also widened by the reach of a meaningless t under quirk Q-4 (mesh_ray_reach below; computed where the
// ray is prepared and carried in its record: the traversal kernel's registers are counted).
__device__ inline float grid_tag(float g) {
    const uint32_t b = (uint32_t)__float_as_int(g);
    return __int_as_float((int)((b & ~31u) | (g < 0 ? 16u : 0u)));
}
// a node word lo | hi << 16 turned so that the face the ray enters through is in the low half (g: a tagged MeshRay::gx..gz)
__device__ inline uint32_t near_far(uint32_t w, float g) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(w, w, (uint32_t)__float_as_int(g));
#else
    return ((uint32_t)__float_as_int(g) & 16u) ? (w >> 16) | (w << 16) : w;
#endif
}
__device__ inline void mesh_ray_grid(MeshRay& r, float4 origin, float4 step, float reach) {
    const float gx = r.idx * step.x, gy = r.idy * step.y, gz = r.idz * step.z;
    const float ox = (r.o.x - origin.x) * r.idx, oy = (r.o.y - origin.y) * r.idy, oz = (r.o.z - origin.z) * r.idz;
    // t = q * g - o_ carries the rounding of both terms: up to ~1.2e-7 of their magnitudes, which is no longer small
    // against the boxes when the ray starts thousands of mesh extents away (a mesh seen from 1e5 units: 4 of 57 k hits
    // were culled, at 1e6 units 1863).  The reference's own slab test rounds just as badly but is not what culls here,
    // so every slab is widened by s = 4e-7 x (largest |q * g| + |o_|) at no cost in the loop: the face the ray ENTERS
    // through gets o_ + s, the one it leaves through o_ - s.
    // Which face that is, is the sign of g, known per ray: the low five bits of g are given up to say so (0 or 16, the
    // amount node_test rotates a node's lo|hi word by, read by v_alignbit straight from g's register: no register and no
    // instruction of the traversal loop spent on it).  That moves g by up to 31 ulp = 3.7e-6 of itself (t by that x 65535 |g|;
    // 32 denormal steps x 65535 < 1e-37 when g is that small), which the widening covers as well.
    float sx = 4.2e-6f * fabsf(gx) * 65535.0f + 4e-7f * fabsf(ox) + 1e-37f, sy = 4.2e-6f * fabsf(gy) * 65535.0f + 4e-7f * fabsf(oy) + 1e-37f,
          sz = 4.2e-6f * fabsf(gz) * 65535.0f + 4e-7f * fabsf(oz) + 1e-37f;
    sx += reach; sy += reach; sz += reach;
    r.gx = grid_tag(gx); r.gy = grid_tag(gy); r.gz = grid_tag(gz);
    r.oxn = ox + sx; r.oxf = ox - sx;
    r.oyn = oy + sy; r.oyf = oy - sy;
    r.ozn = oz + sz; r.ozf = oz - sz;
}
__device__ inline MeshRay mesh_ray_setup(vec3 o, vec3 d, uint32_t quirks, float4 origin, float4 step) {
    MeshRay r;
    r.o = o; r.d = d;
    r.tr = tri_ray_setup(o, d, quirks);
    // A zero (or denormal) direction component must stay usable: 1/0 = inf would turn o * (1/d) into inf and
    // the fused b * (1/d) - o * (1/d) into NaN, culling everything, whereas the reference's division-based slab
    // test (aabb.h:28-31) yields -inf / +inf and passes a box whose slab contains the origin.  A huge finite
    // reciprocal keeps the signs of (b - o) / d, which is all the slab comparison needs.
    r.idx = fabsf(d.x) < 1e-30f ? copysignf(1e30f, d.x) : 1.0f / d.x;
    r.idy = fabsf(d.y) < 1e-30f ? copysignf(1e30f, d.y) : 1.0f / d.y;
    r.idz = fabsf(d.z) < 1e-30f ? copysignf(1e30f, d.z) : 1.0f / d.z;
    // Nothing in mesh_ray_grid may overflow: inf - inf = NaN in a slab term culls the box.  |1/d| is at most 1e30 here, so
    // a mesh more than ~1e8 units wide, or seen from that far, met by a ray with a (near-)zero direction component would
    // (a chain of triangles 7e10 units long lost EVERY hit of the rays parallel to an axis).  Cap the reciprocal on such an
    // axis so that both terms stay below 1e37: the slab's entry and exit still come out as -/+ 1e32 or more, beyond any real
    // t, which is all a (near-)parallel ray needs from them.  Never active below those sizes, so nothing else changes.
    // (idx..idz feed the culling arithmetic only; the wavefront record carries the capped values.)
    {
        const float mx = fmaxf(65535.0f * step.x, fabsf(o.x - origin.x)), my = fmaxf(65535.0f * step.y, fabsf(o.y - origin.y)),
                    mz = fmaxf(65535.0f * step.z, fabsf(o.z - origin.z));
        if (fabsf(r.idx) * mx > 1e37f) r.idx = copysignf(1e37f / mx, r.idx);
        if (fabsf(r.idy) * my > 1e37f) r.idy = copysignf(1e37f / my, r.idy);
        if (fabsf(r.idz) * mz > 1e37f) r.idz = copysignf(1e37f / mz, r.idz);
    }
    r.reach = mesh_ray_reach(o, r.tr.sZ, origin, step, quirks);
    mesh_ray_grid(r, origin, step, r.reach);
    return r;
}
#define HRT_TRAV_DONE 0x7fffffff
#define HRT_TIE_SELF 0x40000000   // TravState::best flag: the hit so far passes triangle.cpp:106-109 against its own t
#define HRT_TIE_OVERFLOW 0xfffffffeu   // TravState::self_order while there is no self-hit: more near-tie contenders than the one note holds
// WorldHit::sub flag of the wavefront pipeline: the traversal met such a tie; the triangle recorded is its own best guess (its t
// is within an ulp or two of the reference's), and world_rec settles it with ref_walk before the hitRecord is built.  (Not in
// the traversal kernel itself: its register budget is counted, tests/test_kernel_resources.py.)
#define HRT_SUB_TIE_UNSETTLED 0x40000000
struct TravState {
    float closest;        // t_max, shrinking
    int best;             // closest accepted triangle so far (mesh-local index) or -1
    // (no member of its own for the near-tie note: while there is no self-hit, `self_tri` holds a triangle whose t lies within
    //  2 ulp of `closest`, or -1 -- see trav_result; the first self-hit overwrites it, and then it is no longer needed)
    // Self-hits (Q-2): a candidate with t < t_min.  In the reference, once one is accepted every later
    // BOX test fails (bvh.cpp:71 gets t_max = rec.t < t_min), so the one in the FIRST lowest-level node of
    // its own tree's depth-first walk wins, whatever its distance (the two triangles of that one node are
    // both tested, bvh.cpp:74-75).  Here: after the first self-hit the interval shrinks to
    // [0, t_min+] -- only boxes that close to the origin can hold another self-hit -- and candidates are
    // ranked by hrt_flat_scene::tri_ref_order.
    bool selfhit;
    uint32_t self_order;
    int self_tri;
    float self_t;
    int sp;
    int cur;              // node index (>= 0), leaf code (< 0) or HRT_TRAV_DONE
};
// t_max as a mesh sees it.  A NaN t_max (the hit so far came from a degenerate primitive: Triangle::hit with two equal
// vertices "hits" every ray with t = NaN) acts like +inf in the reference: aabb.h:34 std::min(t1, NaN) yields t1, and
// triangle.cpp:106-109 compare against NaN * det, which rejects nothing -- exactly what +inf does.  The culling arithmetic
// would otherwise drop a NaN RAY at the root (every slab NaN, exit = fmin(NaN, NaN)), where the reference lets it through
// every box and has the mesh "hit" it.
__device__ inline float mesh_t_max(float t_max) { return t_max == t_max ? t_max : __builtin_huge_valf(); }
template <class M>
__device__ inline void trav_init(TravState& ts, const M& mesh, float t_max) {
    ts.closest = t_max; ts.best = -1;
    ts.selfhit = false; ts.self_order = 0xffffffffu; ts.self_tri = -1; ts.self_t = 0.0f;
    ts.sp = 0;
    ts.cur = mesh.node_count == 0 ? HRT_TRAV_DONE : 0;
}
// Culling interval [t_lo, closest].  With Q-2 the reference accepts triangle hits at any t > 0 as long as
// the boxes of ITS tree pass [t_min, t_max]; those boxes are not the ones of this tree, so cull from 0
// (every t > 0 candidate is then reached) and let accept_box apply the reference's own leaf-level box.
// ts.cur encodings: inner node index in [0, HRT_TRAV_DONE), leaf code < 0, HRT_TRAV_DONE = finished.
__device__ inline bool trav_at_inner(const TravState& ts) { return (unsigned)ts.cur < (unsigned)HRT_TRAV_DONE; }
__device__ inline bool trav_at_leaf(const TravState& ts) { return ts.cur < 0; }
__device__ inline float trav_t_lo(float t_min, uint32_t quirks) { return (quirks & HRT_Q2_TRI_NO_TMIN) ? 0.0f : t_min; }

// Slab test of BOTH child boxes of one 32-byte node against [t_lo, t_hi] (culling only).
__device__ inline void node_test(const uint4& A, const uint4& B, const MeshRay& r, float t_lo, float t_hi,
                                 float& tn0, bool& h0, float& tn1, bool& h1) {
    // per axis: entry = near face, exit = far face (no min / max of the two: the rotation has sorted them)
    const uint32_t ax = near_far(A.x, r.gx), ay = near_far(A.y, r.gy), az = near_far(A.z, r.gz);
    float a0 = fmaf((float)(ax & 0xffffu), r.gx, -r.oxn), a1 = fmaf((float)(ax >> 16), r.gx, -r.oxf);
    float b0 = fmaf((float)(ay & 0xffffu), r.gy, -r.oyn), b1 = fmaf((float)(ay >> 16), r.gy, -r.oyf);
    float c0 = fmaf((float)(az & 0xffffu), r.gz, -r.ozn), c1 = fmaf((float)(az >> 16), r.gz, -r.ozf);
    tn0 = fmaxf(fmaxf(fmaxf(a0, b0), c0), t_lo);
    float tf0 = fminf(fminf(fminf(a1, b1), c1), t_hi);
    const uint32_t bx = near_far(B.x, r.gx), by = near_far(B.y, r.gy), bz = near_far(B.z, r.gz);
    float e0 = fmaf((float)(bx & 0xffffu), r.gx, -r.oxn), e1 = fmaf((float)(bx >> 16), r.gx, -r.oxf);
    float f0 = fmaf((float)(by & 0xffffu), r.gy, -r.oyn), f1 = fmaf((float)(by >> 16), r.gy, -r.oyf);
    float g0 = fmaf((float)(bz & 0xffffu), r.gz, -r.ozn), g1 = fmaf((float)(bz >> 16), r.gz, -r.ozf);
    tn1 = fmaxf(fmaxf(fmaxf(e0, f0), g0), t_lo);
    float tf1 = fminf(fminf(fminf(e1, f1), g1), t_hi);
    h0 = tn0 <= tf0;
    h1 = tn1 <= tf1;
}
// The wavefront pipeline's root filter: would the first traversal step find any child of the root?
// (Exactly the test trav_inner performs on node 0, so filtering changes no result.)
template <class M>
__device__ inline bool root_may_hit(const DScene& sc, const M& mesh, const MeshRay& r, float t_lo, float t_hi) {
    if (mesh.node_count == 0) return false;
    const HRT_CONST_AS uint32_t* n = uniform_table((const uint32_t*)sc.qnodes) + 8ull * mesh.node_first;   // wave-uniform
    uint4 A, B;
    A.x = n[0]; A.y = n[1]; A.z = n[2]; A.w = n[3]; B.x = n[4]; B.y = n[5]; B.z = n[6]; B.w = n[7];
    float tn0, tn1; bool h0, h1;
    node_test(A, B, r, t_lo, t_hi, tn0, h0, tn1, h1);
    return h0 || h1;
}
// grid origin / step of mesh `mi` (wave-uniform)
__device__ inline void mesh_grid(const DScene& sc, int mi, float4& origin, float4& step) {
    const HRT_CONST_AS float* g = uniform_table((const float*)sc.grids) + 8 * mi;
    origin.x = g[0]; origin.y = g[1]; origin.z = g[2]; origin.w = 0.0f;
    step.x = g[4]; step.y = g[5]; step.z = g[6]; step.w = 0.0f;
}

// One inner-node step: tests both children of node ts.cur, descends / pushes / pops.
template <bool STATS>
__device__ inline void trav_inner(const uint4* __restrict__ nodes, const MeshRay& r, TravState& ts, float t_lo, int* stack,
                                  DCounters& cnt, unsigned long long* stp = nullptr) {
    const int cur = ts.cur;
    const uint4 A = nodes[2 * cur + 0];
    const uint4 B = nodes[2 * cur + 1];
#if defined(HRT_TA_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    // experiment (tests/tools/ta_probe.sh): HRT_TA_PROBE more 16-byte loads of the same record per node step, results unused -- the
    // same cache line, no longer dependency chain (the step waits for A and B anyway), only more lane-accesses for the L1's
    // address / tag pipeline: if the kernel's time follows, that pipeline is what bounds it
    {
        const uint4* pp = nodes + 2 * cur;
        uint4 X;
#pragma unroll
        for (int k_ = 0; k_ < HRT_TA_PROBE; ++k_) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(X) : "v"(pp), "n"(16 * (k_ & 1)) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(X.x));
    }
#endif
#if defined(HRT_VALU_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    // experiment (tests/tools/ta_probe.sh): HRT_VALU_PROBE more vector instructions per node step (independent v_fma on two scratch
    // registers): how much of an issue slot's worth of time does the kernel pay for an instruction?
    {
        float p0_ = 1.0f, p1_ = 2.0f;
#pragma unroll
        for (int k_ = 0; k_ < HRT_VALU_PROBE / 2; ++k_) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(p0_)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(p1_)); }
        asm volatile("" :: "v"(p0_), "v"(p1_));
    }
#endif
#ifdef HRT_STEP_PROFILE      // experiments: where a node step's cycles go (s_memtime after forced waits; see tests/tools/step_profile_run.py)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    { const unsigned long long n_ = __builtin_readcyclecounter(); stp[0] += n_ - stp[3]; stp[3] = n_; stp[6] += 1; }
#endif
    if (STATS) cnt.box_tests += 2;
    float tn0, tn1; bool h0, h1;
    node_test(A, B, r, t_lo, ts.closest, tn0, h0, tn1, h1);
    const int ch0 = (int)A.w, ch1 = (int)B.w;
#ifdef HRT_STEP_PROFILE
    {   // (the compare results must exist before the clock is read)
        asm volatile("" :: "v"(tn0), "v"(tn1), "s"(__ballot(h0)), "s"(__ballot(h1)));
        const unsigned long long n_ = __builtin_readcyclecounter(); stp[1] += n_ - stp[3]; stp[3] = n_;
    }
#endif
    if (h0 && h1) {
        const bool swap = tn1 < tn0;
        const int nearc = swap ? ch1 : ch0, farc = swap ? ch0 : ch1;
        stack[ts.sp * HRT_BLOCK] = farc; ++ts.sp;
        ts.cur = nearc;
    } else if (h0) ts.cur = ch0;
    else if (h1) ts.cur = ch1;
    else if (ts.sp > 0) { --ts.sp; ts.cur = stack[ts.sp * HRT_BLOCK]; }
    else ts.cur = HRT_TRAV_DONE;
#ifdef HRT_STEP_PROFILE
    {
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(ts.cur) : "memory");
        const unsigned long long n_ = __builtin_readcyclecounter(); stp[2] += n_ - stp[3]; stp[3] = n_;
    }
#endif
}

// One leaf: every triangle of leaf code ts.cur through the exact ITriangle test, then pop.
template <bool STATS>
__device__ inline void trav_leaf(const float4* __restrict__ tpos, const float4* __restrict__ tbox, const MeshRay& r, TravState& ts,
                                 float t_min, uint32_t quirks, int* stack, DCounters& cnt) {
    const uint32_t enc = (uint32_t)(~ts.cur);
    const uint32_t first = enc >> 3, count = (enc & 7u) + 1u;
    for (uint32_t k = 0; k < count; ++k) {
        const uint32_t ti = first + k;
        const float4 q0 = tpos[3 * ti + 0], q1 = tpos[3 * ti + 1], q2 = tpos[3 * ti + 2];
        if (STATS) cnt.tri_tests += 1;
        TriEval ev;
        if (!tri_eval(r.tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev)) continue;
        // triangle.cpp:106-109
        const float lim = ts.closest * ev.det;
        bool farther = false;
        if (ev.det < 0) { if (ev.tScaled >= 0) continue; farther = ev.tScaled < lim; }
        else if (ev.det > 0) { if (ev.tScaled <= 0) continue; farther = ev.tScaled > lim; }
        // `farther` rejects -- except for a rounding tie with the hit so far, see "Ties" below
        if (farther && (ts.best < 0 || ts.selfhit || fabsf(ev.tScaled - lim) > 1e-6f * fabsf(lim))) continue;
        const float invDet = 1 / ev.det;
        const float t = ev.tScaled * invDet;
        // within 2 ulp of the hit so far (both positive floats: their bit patterns are ordered like their values)?
        const bool near = ts.best >= 0 && (unsigned)(__float_as_int(t) - __float_as_int(ts.closest) + 2) <= 4u;
        if (farther && !near) continue;
        if (!(quirks & HRT_Q2_TRI_NO_TMIN) && t < t_min) continue;
        const float4 bmn = tbox[2 * ti], bmx = tbox[2 * ti + 1];
        // accept_box costs six IEEE divisions and almost always passes.  Shortcut with the SAME outcome: if t is
        // clear of t_min, closest >= t, and the hit point o + t d lies inside the box by a margin far above the
        // rounding of (box - o) / d, then on every axis the slab interval strictly contains t, so after clipping
        // to [t_min, closest] the interval is non-empty and aabb.h:35 cannot reject.  Anything closer to a face,
        // t near t_min, or non-finite takes the exact test.
        {
            const float hx = r.o.x + t * r.d.x, hy = r.o.y + t * r.d.y, hz = r.o.z + t * r.d.z;
            const float mgx = 1e-4f * (1.0f + fabsf(hx) + fabsf(r.o.x)), mgy = 1e-4f * (1.0f + fabsf(hy) + fabsf(r.o.y)),
                        mgz = 1e-4f * (1.0f + fabsf(hz) + fabsf(r.o.z));
            const bool inside = t > 1.001f * t_min && ts.closest >= t && hx - bmn.x > mgx && bmx.x - hx > mgx && hy - bmn.y > mgy &&
                                bmx.y - hy > mgy && hz - bmn.z > mgz && bmx.z - hz > mgz;
            if (!inside && !accept_box(bmn, bmx, r.o, r.d, t_min, ts.closest)) continue;
        }
        if (t < t_min) {
            const uint32_t ord = (uint32_t)__float_as_int(bmn.w);
            bool take;
            if (!ts.selfhit) take = true;
            else if ((ord >> 1) != (ts.self_order >> 1)) take = (ord >> 1) < (ts.self_order >> 1);
            else if (ord & 1u) {
                // this is `right`, the kept one is `left`: bvh.cpp:75 hands `right` t_max = left's t, and `right` stays
                // unless triangle.cpp:106-109 find it strictly beyond that
                const float lim_l = ts.self_t * ev.det;
                take = ev.det < 0 ? !(ev.tScaled < lim_l) : !(ev.tScaled > lim_l);
            } else {
                // this is `left`, the kept one is `right`, which the reference tests AFTER this one with t_max = this t
                // (bvh.cpp:75): `right` stays unless triangle.cpp:106-109 find it strictly beyond that -- a comparison of ITS
                // tScaled with fl(t * det), which it can pass with a t one ulp LARGER than this one (a ray leaving the shared edge
                // of a front- and a back-facing triangle: both report t ~ 1e-6, tests/tools/gpu_fuzz_meshes.py).  The kept
                // triangle's tScaled and det are not kept: evaluate it again (a cold path: two self-hits in one node).
                const int rt = ts.self_tri & ~HRT_TIE_SELF;
                const float4 r0 = tpos[3 * rt + 0], r1 = tpos[3 * rt + 1], r2 = tpos[3 * rt + 2];
                TriEval er;
                take = true;
                if (tri_eval(r.tr, vec3(r0.x, r0.y, r0.z), vec3(r1.x, r1.y, r1.z), vec3(r2.x, r2.y, r2.z), er)) {
                    const float lim_r = t * er.det;
                    take = er.det < 0 ? (er.tScaled < lim_r) : (er.tScaled > lim_r);
                }
            }
            if (take) {
                const float own = t * ev.det;
                const bool self_now = ev.det < 0 ? !(ev.tScaled < own) : !(ev.tScaled > own);
                ts.self_order = ord; ts.self_tri = (int)ti | (self_now ? HRT_TIE_SELF : 0); ts.self_t = t;
            }
            if (!ts.selfhit) { ts.selfhit = true; ts.closest = t_min * 1.0001f; }
        } else if (!ts.selfhit) {
            // Ties and near-ties.  triangle.cpp:106-109 keep a candidate X unless tScaled_X is strictly beyond t_best * det_X
            // in fp32, and the reference meets the triangles in the fixed left-then-right order of ITS tree (bvh.cpp:74-75,
            // ranked by tri_ref_order), this tree in another.  Between two hits whose t lie within an ulp or two (a shared
            // edge, duplicated or coplanar faces, far-away origins) that comparison is not a total order -- the later one
            // can be turned down although its t is an ulp smaller, or taken although it is equal -- so the winner depends on
            // who is met first.  Here the walk keeps the closer one as usual and remembers the other (in `self_tri`, unused until a self-hit); trav_result
            // replays the pair in the reference's order.
            // A THIRD contender (three hits within an ulp or two: a vertex shared by faces seen edge-on) cannot be settled from
            // one note: the reference's fold over them is not transitive.  Such a ray walks the reference's tree (trav_tie_overflow).
            if (farther) {
                if (ts.self_tri >= 0 && ts.self_tri != (int)ti && t == t) ts.self_order = HRT_TIE_OVERFLOW;
                ts.self_tri = (int)ti;
            } else {
                if (near) {
                    // (not for NaN: a NaN ray passes every box and "hits" every triangle with t = NaN -- all of them "near" each
                    //  other; the record is NaN whichever stays, and a walk of the whole reference tree per such ray is 8 ms)
                    if (ts.self_tri >= 0 && ts.self_tri != ts.best && t == t) ts.self_order = HRT_TIE_OVERFLOW;
                    ts.self_tri = ts.best;
                }
                ts.closest = t;
                ts.best = (int)ti;
            }
        }
    }
    if (ts.sp > 0) { --ts.sp; ts.cur = stack[ts.sp * HRT_BLOCK]; }
    else ts.cur = HRT_TRAV_DONE;
}
// More than two hits within an ulp or two of each other were met (and no self-hit, which would decide anyway): the result must
// come from ref_walk, the reference's own walk.
__device__ inline bool trav_tie_overflow(const TravState& ts) { return !ts.selfhit && ts.self_order == HRT_TIE_OVERFLOW; }
// Result of a finished traversal: winning triangle (or -1) and its t.  A remembered near-tie (ts.self_tri without a self-hit) is settled the way the
// reference's walk would have: F = the one of the pair its tree meets first, S = the other; F stands unless S, tested with
// t_max = t_F (bvh.cpp:75 / the running closest), passes triangle.cpp:106-109 and its leaf-level box (aabb.h:26-39).
__device__ inline int trav_result(const TravState& ts, const float4* __restrict__ tpos, const float4* __restrict__ tbox, const MeshRay& r,
                                  float t_min, float& t_out) {
    if (ts.selfhit) { t_out = ts.self_t; return ts.self_tri & ~HRT_TIE_SELF; }
    t_out = ts.closest;
    const int alt = ts.self_tri;
    if (ts.best < 0 || alt < 0 || alt == ts.best) return ts.best;
    const uint32_t ord_b = (uint32_t)__float_as_int(tbox[2 * ts.best].w), ord_a = (uint32_t)__float_as_int(tbox[2 * alt].w);
    const int F = ord_a < ord_b ? alt : ts.best, S = ord_a < ord_b ? ts.best : alt;
    // (one evaluation body run twice, not two bodies: the registers of k_wf_ext are counted, tests/test_kernel_resources.py)
    TriEval eS;
    float tF = 0.0f;
#if defined(__HIPCC__)
#pragma clang loop unroll(disable)
#endif
    for (int k = 0; k < 2; ++k) {
        const int T = k == 0 ? F : S;
        const float4 q0 = tpos[3 * T + 0], q1 = tpos[3 * T + 1], q2 = tpos[3 * T + 2];
        if (!tri_eval(r.tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), eS)) return ts.best;
        if (k == 0) tF = eS.tScaled * (1 / eS.det);
    }
    const float tS = eS.tScaled * (1 / eS.det);
    // a stale note (the pair it was about has been overtaken by a clearly closer hit): nothing to settle
    if ((unsigned)(__float_as_int(tS) - __float_as_int(tF) + 2) > 4u) return ts.best;
    const float lim = tF * eS.det;
    const bool farther = eS.det < 0 ? eS.tScaled < lim : eS.tScaled > lim;
    const bool s_wins = !farther && accept_box(tbox[2 * S], tbox[2 * S + 1], r.o, r.d, t_min, tF);
    t_out = s_wins ? tS : tF;
    return s_wins ? S : F;
}

// Whole traversal for one lane (megakernel / test kernels).
template <bool STATS>
__device__ inline int bvh_traverse(const DScene& sc, int mi /* mesh index */, vec3 o, vec3 d, float t_min, float t_max,
                                   uint32_t quirks, int* stack /* + threadIdx.x */, float& t_out, DCounters& cnt) {
    const auto& mesh = uniform_table(sc.meshes)[mi];
    const uint4* nodes = sc.qnodes + 2ull * mesh.node_first;
    const float4* tpos = sc.tri_pos + 3ull * mesh.tri_first;
    const float4* tbox = sc.tri_box + 2ull * mesh.tri_first;
    float4 grid_o, grid_s;
    mesh_grid(sc, mi, grid_o, grid_s);
    const MeshRay r = mesh_ray_setup(o, d, quirks, grid_o, grid_s);
    if (q4_risky(r.tr, d, quirks, sc.q4_route_a2)) {
        const HRT_CONST_AS uint32_t* rm = uniform_table((const uint32_t*)sc.rmesh) + 4 * mi;
        if (sc.ref_fold_all && d.x != 0.0f && d.y != 0.0f && d.z != 0.0f)
            return ref_fold<STATS>(sc.rtris + 3ull * rm[2], tbox, mesh.tri_count, o, d, r.tr, t_min, mesh_t_max(t_max), quirks, t_out, cnt);
        return ref_walk<STATS>(sc.rnodes + 2ull * rm[0], sc.rtris + 3ull * rm[2], rm[1], o, d, r.tr, t_min, mesh_t_max(t_max), quirks, t_out, cnt);
    }
    const float t_lo = trav_t_lo(t_min, quirks);
    TravState ts;
    trav_init(ts, mesh, mesh_t_max(t_max));
    // while-while: lanes that reach a leaf wait at the end of the inner loop, so the (expensive) leaf code
    // runs with many lanes at once instead of once per node step for the few lanes that happen to be at a leaf
    while (ts.cur != HRT_TRAV_DONE) {
        while (trav_at_inner(ts)) trav_inner<STATS>(nodes, r, ts, t_lo, stack, cnt);
        if (trav_at_leaf(ts)) trav_leaf<STATS>(tpos, tbox, r, ts, t_min, quirks, stack, cnt);
    }
    if (trav_tie_overflow(ts)) {
        const HRT_CONST_AS uint32_t* rm = uniform_table((const uint32_t*)sc.rmesh) + 4 * mi;
        return ref_walk<STATS>(sc.rnodes + 2ull * rm[0], sc.rtris + 3ull * rm[2], rm[1], o, d, r.tr, t_min, mesh_t_max(t_max), quirks, t_out, cnt);
    }
    return trav_result(ts, tpos, tbox, r, t_min, t_out);
}

// triangle.cpp:111-128 for the winning triangle
__device__ inline void tri_rec(const DScene& sc, const hrt_mesh& mesh, int tri, vec3 o, vec3 d, uint32_t quirks, DRec& rec) {
    HRT_BOUNDS(0, tri, mesh.tri_count);
    const float4* tpos = sc.tri_pos + 3ull * mesh.tri_first;
    const float4* tattr = sc.tri_attr + 4ull * mesh.tri_first;
    const TriRay tr = tri_ray_setup(o, d, quirks);
    const float4 q0 = tpos[3 * tri + 0], q1 = tpos[3 * tri + 1], q2 = tpos[3 * tri + 2];
    TriEval ev;
    tri_eval(tr, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), ev);
    float invDet = 1 / ev.det;
    float b0 = ev.e0 * invDet;
    float b1 = ev.e1 * invDet;
    float b2 = ev.e2 * invDet;
    float t = ev.tScaled * invDet;
    rec.t = t;
    rec.p = o + (t * d);
    const float4 a0 = tattr[4 * tri + 0], a1 = tattr[4 * tri + 1], a2 = tattr[4 * tri + 2], a3 = tattr[4 * tri + 3];
    vec3 normal = b0 * vec3(a0.x, a0.y, a0.z) + b1 * vec3(a1.x, a1.y, a1.z) + b2 * vec3(a2.x, a2.y, a2.z);
    vec2 uv = b0 * vec2(a0.w, a1.w) + b1 * vec2(a2.w, a3.x) + b2 * vec2(a3.y, a3.z);
    rec.normal = normal;
    rec.u = uv.x;
    rec.v = uv.y;
    if (quirks & HRT_Q3_TRI_NO_FACE) rec.frontFace = true;
    else set_face_normal(rec, d, normal);
}

// ------------------------------------------------------------------ world->hit (hittableList.cpp:4-21 over scene.cpp:376-379)
// Stale frontFace (hittableList.cpp:6-16 with triangle.cpp:118-128, quirk Q-3): HittableList::hit hands every object the same
// tempRec, and ITriangle::hit never writes frontFace -- so the hit of a mesh that stands in the list WITHOUT a wrapper (a wrapper's
// setFaceNormal would write it, translate.cpp:16) carries the flag of the previous successful object of that walk which does write
// it: any object but another such mesh.  Only Dielectric::scatter reads the flag (material.h:207), so this is tracked only in
// scenes that have such a mesh with such a material (DScene::stale_ff): the winner's `sub` gets HRT_SUB_WRAPPERLESS, and
// (s_prim, s_sub, s_t) name the hit whose flag it inherits, s_prim = -1 when there was none (the reference then reads an
// uninitialised bool; defined `true` here and in the oracle).
#define HRT_SUB_WRAPPERLESS 0x20000000
// ... and the wavefront pipeline, which has resolved the source's flag in a stage of its own (k_wf_stale), says "it is false" here:
#define HRT_SUB_STALE_BACK 0x10000000
struct WorldHit { int prim; int sub; float t; int s_prim; int s_sub; float s_t; };  // sub = triangle index (mesh) or box side
__device__ inline bool stale_ff_tracked(const DScene& sc, uint32_t quirks) { return sc.stale_ff && (quirks & HRT_Q3_TRI_NO_FACE); }

template <bool STATS>
__device__ inline WorldHit world_hit(const DScene& sc, vec3 o, vec3 d, float t_min, float t_max, uint32_t quirks,
                                     const rng_ctx& ctx, int* stack, DCounters& cnt) {
    WorldHit wh; wh.prim = -1; wh.sub = -1; wh.t = t_max; wh.s_prim = -1; wh.s_sub = -1; wh.s_t = 0.0f;
    float closest = t_max;
    const bool track = stale_ff_tracked(sc, quirks);
    for (int i = 0; i < sc.n_prims; ++i) {
        const auto& pr = uniform_table(sc.prims)[i];
        vec3 lo = o, ld = d;
        for (int k = 0; k < pr.n_xforms; ++k) xf_apply(pr.xf[k], lo, ld, quirks);
        float t; int sub = -1; bool hit;
        const int kind = pr.kind;
        if (kind == HRT_PRIM_MESH) {
            sub = bvh_traverse<STATS>(sc, pr.mesh, lo, ld, t_min, closest, quirks, stack, t, cnt);
            hit = sub >= 0;
        } else if (kind == HRT_PRIM_SPHERE) {
            hit = sphere_hit(pr.p, lo, ld, t_min, closest, t);
        } else if (kind == HRT_PRIM_BOX) {
            hit = box_hit(pr.p, lo, ld, t_min, closest, t, sub);
        } else if (kind == HRT_PRIM_MEDIUM) {
            hit = medium_hit(pr, (uint32_t)i, lo, ld, t_min, closest, ctx, t);
        } else if (kind == HRT_PRIM_TRIANGLE) {
            hit = triangle_hit(pr.p, lo, ld, t_min, closest, t);
        } else {
            hit = rect_hit(rect_axis(kind), pr.p, lo, ld, t_min, closest, t);
        }
        if (hit) {
            if (track) {
                if (kind == HRT_PRIM_MESH && pr.n_xforms == 0) sub |= HRT_SUB_WRAPPERLESS;      // inherits the flag: the source stays
                else { wh.s_prim = i; wh.s_sub = sub; wh.s_t = t; }
            }
            closest = t; wh.prim = i; wh.sub = sub; wh.t = t;
        }
    }
    return wh;
}

// The analytic (non-mesh) part of the world list walk for prims [p0, p1): the body of the loop at
// hittableList.cpp:12-19 with closest-so-far carried in and out.  Mesh prims are skipped (the wavefront
// pipeline handles them between two calls of this function, in list order).
__device__ inline void prims_range_hit(const DScene& sc, int p0, int p1, vec3 o, vec3 d, float t_min, uint32_t quirks,
                                       const rng_ctx& ctx, float& closest, int& prim, int& sub) {
    for (int i = p0; i < p1; ++i) {
        const auto& pr = uniform_table(sc.prims)[i];
        const int kind = pr.kind;
        if (kind == HRT_PRIM_MESH) continue;
        vec3 lo = o, ld = d;
        for (int k = 0; k < pr.n_xforms; ++k) xf_apply(pr.xf[k], lo, ld, quirks);
        float t; int sb = -1; bool hit;
        if (kind == HRT_PRIM_SPHERE) hit = sphere_hit(pr.p, lo, ld, t_min, closest, t);
        else if (kind == HRT_PRIM_BOX) hit = box_hit(pr.p, lo, ld, t_min, closest, t, sb);
        else if (kind == HRT_PRIM_MEDIUM) hit = medium_hit(pr, (uint32_t)i, lo, ld, t_min, closest, ctx, t);
        else if (kind == HRT_PRIM_TRIANGLE) hit = triangle_hit(pr.p, lo, ld, t_min, closest, t);
        else hit = rect_hit(rect_axis(kind), pr.p, lo, ld, t_min, closest, t);
        if (hit) { closest = t; prim = i; sub = sb; }
    }
}

// Rebuilds the winner's hitRecord exactly as the reference's call chain does.
// stale_back: the winner is a mesh hit marked HRT_SUB_WRAPPERLESS and the frontFace it inherits is `false`.
__device__ inline void world_rec(const DScene& sc, const WorldHit& wh, vec3 o, vec3 d, uint32_t quirks, float t_min_for_ties, DRec& rec,
                                 bool stale_back = false) {
    int prim_i = wh.prim;
    HRT_BOUNDS(1, prim_i, sc.n_prims);
    const hrt_prim& pr = sc.lprims[prim_i];
    vec3 lo = o, ld = d;
    const int n = pr.n_xforms;
    // The wrapper chain, outermost first; d_k = the direction wrapper k handed to its child.  Written as NESTED ifs on
    // purpose: the flat form (`#pragma unroll for k: if (k < n) xf_apply(xf[k])`, four independent predicated blocks) is
    // miscompiled by hipcc 7.2 for gfx950 inside k_wf_shade / k_wf_tail -- a per-lane chain of exactly three wrappers whose
    // innermost one is a rotate_y came back with a wrong rec.p.  Root-caused in round 2 (tests/tools/repro_wrapper_chain.md):
    // SimplifyCFG's sinking of common instructions plus the SLP vectorizer on the unrolled blocks; either pass off, -O1, or
    // this nested shape (block k + 1 inside block k's condition) is right.  The flat shape is kept compilable behind
    // -DHRT_XF_FLAT for the reproducer; tests/test_gpu_scenes.py::test_wrapper_chains_of_every_length pins the result.
    vec3 d0 = d, d1 = d, d2 = d, d3 = d;
    static_assert(HRT_MAX_XFORMS == 4, "world_rec spells the wrapper chain out");
#ifdef HRT_XF_FLAT   // the shape that came out wrong (tests/tools/repro_wrapper_chain.md): kept compilable for the reproducer only
#ifdef HRT_XF_FLAT_INIT
    vec3 dirs[HRT_MAX_XFORMS] = {d, d, d, d};
#else
    vec3 dirs[HRT_MAX_XFORMS];
#endif
#pragma unroll
    for (int k = 0; k < HRT_MAX_XFORMS; ++k) {
        if (k < n) { xf_apply(pr.xf[k], lo, ld, quirks); dirs[k] = ld; }
    }
    if (false) {
#else
    if (n > 0) {
#endif
        xf_apply(pr.xf[0], lo, ld, quirks); d0 = ld;
        if (n > 1) {
            xf_apply(pr.xf[1], lo, ld, quirks); d1 = ld;
            if (n > 2) {
                xf_apply(pr.xf[2], lo, ld, quirks); d2 = ld;
                if (n > 3) { xf_apply(pr.xf[3], lo, ld, quirks); d3 = ld; }
            }
        }
    }
    rec.mat = pr.material;
    rec.frontFace = true;
    const int kind = pr.kind;
    if (kind == HRT_PRIM_MESH) {
        int tri = sub_tri(wh.sub);
        int mesh_i = pr.mesh;
        HRT_BOUNDS(4, mesh_i, sc.n_meshes);
        if (wh.sub & HRT_SUB_TIE_UNSETTLED) {
            // three or more hits within an ulp or two of each other: the reference's own walk decides (from a t_max just beyond
            // them: whatever lay clearly nearer would have won already)
            const uint32_t* rm = (const uint32_t*)sc.rmesh + 4 * mesh_i;      // (a per-lane index: plain loads)
            DCounters none; none.box_tests = 0; none.tri_tests = 0;
            float t_settled;
            const int exact = ref_walk<false>(sc.rnodes + 2ull * rm[0], sc.rtris + 3ull * rm[2], rm[1], lo, ld, tri_ray_setup(lo, ld, quirks), t_min_for_ties,
                                              wh.t + fabsf(wh.t) * 4e-6f, quirks, t_settled, none);
            if (exact >= 0) tri = exact;
        }
        tri_rec(sc, sc.lmeshes[mesh_i], tri, lo, ld, quirks, rec);
        if (stale_back) rec.frontFace = false;      // (only ever set for a wrapper-less mesh under Q-3: nothing below rewrites it)
    }
    else if (kind == HRT_PRIM_SPHERE) sphere_rec(pr.p, lo, ld, wh.t, rec);
    else if (kind == HRT_PRIM_BOX) box_rec(pr.p, lo, ld, wh.t, wh.sub, rec);
    else if (kind == HRT_PRIM_MEDIUM) {  // constantMedium.cpp:30-36
        rec.t = wh.t; rec.p = lo + (wh.t * ld); rec.normal = vec3(1, 0, 0); rec.frontFace = true; rec.u = 0.0f; rec.v = 0.0f;
    } else if (kind == HRT_PRIM_TRIANGLE) triangle_rec(pr.p, lo, ld, rec);
    else rect_rec(rect_axis(kind), pr.p, lo, ld, wh.t, rec);
#ifdef HRT_XF_FLAT
#pragma unroll
    for (int k = HRT_MAX_XFORMS - 1; k >= 0; --k) {
        if (k < n) xf_unapply(pr.xf[k], rec, dirs[k]);
    }
    if (false) {
#else
    if (n > 0) {   // innermost wrapper first, nested for the same reason
#endif
        if (n > 1) {
            if (n > 2) {
                if (n > 3) xf_unapply(pr.xf[3], rec, d3);
                xf_unapply(pr.xf[2], rec, d2);
            }
            xf_unapply(pr.xf[1], rec, d1);
        }
        xf_unapply(pr.xf[0], rec, d0);
    }
}

// The hitRecord main.cpp:46 gets back for the winner `wh` (wh.prim >= 0).  A wrapper-less mesh hit with a source for its
// frontFace: the source's record first, for that one flag (ONE world_rec body run twice by those lanes, not two bodies: the
// registers of k_wf_shade are counted).
__device__ inline void hit_record(const DScene& sc, const WorldHit& wh, vec3 o, vec3 d, uint32_t quirks, float t_min_for_ties, DRec& rec) {
    int prim_i = wh.prim;
    HRT_BOUNDS(1, prim_i, sc.n_prims);
    const bool wrapperless = sc.lprims[prim_i].kind == HRT_PRIM_MESH && (wh.sub & HRT_SUB_WRAPPERLESS);
    bool stale_back = wrapperless && (wh.sub & HRT_SUB_STALE_BACK);
    const bool inherits = wrapperless && !stale_back && wh.s_prim >= 0;
    WorldHit cur = wh;
    if (inherits) { cur.prim = wh.s_prim; cur.sub = wh.s_sub; cur.t = wh.s_t; }
#if defined(__HIPCC__)
#pragma clang loop unroll(disable)
#endif
    for (int pass = inherits ? 0 : 1; pass < 2; ++pass) {
        world_rec(sc, cur, o, d, quirks, t_min_for_ties, rec, stale_back);
        if (pass == 0) { stale_back = !rec.frontFace; cur = wh; }
    }
}

// ------------------------------------------------------------------ textures (texture.cpp)
__device__ inline vec3 tex_leaf(const DScene& sc, const hrt_texture& t, float u, float v) {
    if (t.kind == HRT_TEX_SOLID) return vec3(t.c[0], t.c[1], t.c[2]);
    if (t.width == 0 || t.height == 0) return vec3(0, 1, 1);  // texture.cpp:56-57,79-80
    if (t.kind == HRT_TEX_IMAGE) {  // texture.cpp:53-74
        u = gclamp(u, 0.0f, 1.0f);
        v = 1.0f - gclamp(v, 0.0f, 1.0f);
        int i = texel_index(u * t.width);
        int j = texel_index(v * t.height);
        if (i >= t.width) i = t.width - 1;
        if (j >= t.height) j = t.height - 1;
        const float colourScale = 1.0f / 255.0f;
        const uint8_t* px = sc.texels_u8 + t.offset + (size_t)j * (3 * t.width) + (size_t)i * 3;
        return vec3(colourScale * px[0], colourScale * px[1], colourScale * px[2]);
    }
    // HRT_TEX_ENV: texture.cpp:76-97
    u = gclamp(u, 0.0f, 1.0f);
    v = gclamp(v, 0.0f, 1.0f);
    int i = texel_index((u * (t.width - 1)) + 0.5f);
    int j = texel_index((v * (t.height - 1)) + 0.5f);
    const float* px = sc.texels_f32 + t.offset + ((size_t)j * t.width + i) * t.channels;
    return vec3(px[0], px[1], px[2]);
}
__device__ inline vec3 tex_value(const DScene& sc, int tex, float u, float v, vec3 p) {
    // CheckeredTexture (texture.cpp:17-28) may nest; bounded walk instead of recursion
    for (int depth = 0; depth < 4; ++depth) {
        HRT_BOUNDS(3, tex, sc.n_texs);
        const hrt_texture& t = sc.ltexs[tex];
        if (t.kind != HRT_TEX_CHECKER) return tex_leaf(sc, t, u, v);
        float sines = gsin_wide(10 * p.x) * gsin_wide(10 * p.y) * gsin_wide(10 * p.z);
        tex = (sines < 0) ? t.odd : t.even;
    }
    return vec3(0, 1, 1);
}
__device__ inline vec3 matvec3_value(const DScene& sc, const hrt_matvec3& m, float u, float v, vec3 p) {
    if (m.tex < 0) return vec3(m.c[0], m.c[1], m.c[2]);
    return tex_value(sc, m.tex, u, v, p);
}
__device__ inline float matscalar_value(const DScene& sc, const hrt_matscalar& m, float u, float v, vec3 p) {
    if (m.tex < 0) return m.c;
    return length(tex_value(sc, m.tex, u, v, p));
}

// ------------------------------------------------------------------ Material::emitted / scatter (material.h, material.cpp)
// Returns false when the path ends (DiffuseLight, absorbed Metal).
__device__ inline bool material_scatter(const DScene& sc, const DRec& rec, vec3 rin_d, const rng_ctx& ctx, vec3& emitted,
                                        vec3& attenuation, vec3& so, vec3& sd) {
    int mat_i = rec.mat;
    HRT_BOUNDS(2, mat_i, sc.n_mats);
    const hrt_material& m = sc.lmats[mat_i];
    int kind = m.kind;
    emitted = vec3(0.0f);
    if (kind == HRT_MAT_DIFFUSE_LIGHT) {  // material.h:96-104
        emitted = matvec3_value(sc, m.albedo, rec.u, rec.v, rec.p) * matscalar_value(sc, m.s0, rec.u, rec.v, rec.p);
        return false;
    }
    so = rec.p;
    if (kind == HRT_MAT_ISOTROPIC) {  // material.h:79-85
        sd = ball_rand(ctx);
        attenuation = matvec3_value(sc, m.albedo, rec.u, rec.v, rec.p);
        return true;
    }
    const u32x4 dr = rng_draw(ctx, RNG_SCATTER, 0);
    const vec3 sph = spherical_rand(dr.x, dr.y);
    if (kind == HRT_MAT_PBR) {  // material.cpp:18-28
        bool metal = length(tex_value(sc, m.mix_tex, rec.u, rec.v, rec.p)) > 0.5f;
        kind = metal ? HRT_MAT_METAL : HRT_MAT_LAMBERTIAN;
    }
    if (kind == HRT_MAT_LAMBERTIAN || kind == HRT_MAT_UVTEST) {  // material.h:137-153, 116-129
        vec3 scatterDirection = rec.normal + sph;
        if (near_zero(scatterDirection)) scatterDirection = rec.normal;
        sd = scatterDirection;
        attenuation = (kind == HRT_MAT_UVTEST) ? rec.normal : matvec3_value(sc, m.albedo, rec.u, rec.v, rec.p);
        return true;
    }
    if (kind == HRT_MAT_METAL) {  // material.h:166-177
        vec3 nn = normalize(rec.normal);
        vec3 reflected = reflect(normalize(rin_d), nn);
        float roughness = fabsf(matscalar_value(sc, m.s0, rec.u, rec.v, rec.p));
        roughness = roughness < 1 ? roughness : 1;
        sd = reflected + roughness * sph + vec3(1.1920928955078125e-7f);
        attenuation = matvec3_value(sc, m.albedo, rec.u, rec.v, rec.p);
        return dot(sd, nn) > 0;
    }
    // HRT_MAT_DIELECTRIC: material.h:204-229, 236-241
    attenuation = vec3(1, 1, 1);
    float irv = matscalar_value(sc, m.s0, rec.u, rec.v, rec.p);
    float refractionRatio = rec.frontFace ? (1.0f / irv) : irv;
    vec3 unitDirection = normalize(rin_d);
    double cosTheta = gmin(dot(-unitDirection, rec.normal), 1.0f);
    double sinTheta = sqrt(1.0 - cosTheta * cosTheta);
    bool cannot_refract = refractionRatio * sinTheta > 1.0;
    double r0 = (1 - refractionRatio) / (1 + refractionRatio);
    r0 = r0 * r0;
    double ref = r0 + (1 - r0) * pow5(1 - cosTheta);
    vec3 direction;
    if (cannot_refract || ref > u01d(dr.z, dr.w)) direction = reflect(unitDirection, rec.normal);
    else direction = refract(unitDirection, rec.normal, refractionRatio);
    sd = direction + matscalar_value(sc, m.s1, rec.u, rec.v, rec.p) * sph;
    return true;
}

// main.cpp:47-58
__device__ inline vec3 background_value(const DScene& sc, vec3 d) {
    const float pi = 3.14159265358979323846264338327950288f;
    vec3 nD = normalize(d);
    float phi = gatan2(nD.z, nD.x);
    float theta = gacos(nD.y);
    float u = phi / (2 * pi) + 0.5f;
    float v = theta / pi;
    return tex_value(sc, sc.background_tex, u, v, vec3(0.0f));
}

// ------------------------------------------------------------------ render() / rayColour() pieces (main.cpp)
struct PathState {
    vec3 o, d;          // current ray
    vec3 atten, result; // currentAttenuation, result (main.cpp:40-41)
    int bounce;
};
struct PathCounters {
    unsigned rays, samples, mesh_hits, env_lookups;
    DCounters bvh;
};

// main.cpp:115-123 + Camera::getRay (camera.h:29-39): starts sample `ctx.sample` of pixel (px, py).
__device__ inline void path_begin(const hrt_camera& cam, const hrt_params& pr, int px, int py, rng_ctx& ctx, PathState& ps) {
    const vec3 c_origin(cam.origin[0], cam.origin[1], cam.origin[2]);
    const vec3 c_llc(cam.lower_left[0], cam.lower_left[1], cam.lower_left[2]);
    const vec3 c_hor(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    const vec3 c_ver(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    ctx.bounce = 0;
    const u32x4 j = rng_draw(ctx, RNG_JITTER, 0);
    const int x = px;                 // main.cpp:115 (pIdx % W)
    const int y = pr.height - py;     // main.cpp:116 (H - pIdx / W), Q-10
    const float u = ((float)x + linear_rand(j.x, 0.0f, 1.0f)) / (pr.width - 1);
    const float v = ((float)y + linear_rand(j.y, 0.0f, 1.0f)) / (pr.height - 1);
    vec3 offset(0.0f);                                   // camera.h:34-35: the reference's lens offset is 0 ...
    if (pr.flags & HRT_FLAG_THIN_LENS) {                 // ... unless the commented-out circularRand(lensRadius) is asked back
        const u32x4 l = rng_draw(ctx, RNG_LENS, 0);
        float rx, ry;
        circular_rand(l.x, cam.lens_radius, rx, ry);
        offset = vec3(cam.lens_u[0], cam.lens_u[1], cam.lens_u[2]) * rx + vec3(cam.lens_v[0], cam.lens_v[1], cam.lens_v[2]) * ry;
    }
    ps.o = c_origin + offset;
    ps.d = c_llc + u * c_hor + v * c_ver - c_origin - offset;
    ps.atten = vec3(1.0f); ps.result = vec3(0.0f); ps.bounce = 0;
}

// The part of one iteration of main.cpp:43-76 that follows world->hit: background on a miss
// (main.cpp:47-59), else emitted + scatter (main.cpp:62-75).  Returns true when the path has ended
// (miss, emitter / absorbed, or MAX_DEPTH segments traced).
template <bool STATS>
__device__ inline bool path_shade(const DScene& sc, const hrt_params& pr, const rng_ctx& ctx, PathState& ps, const WorldHit& wh,
                                  PathCounters& pc) {
    if (wh.prim < 0) {
        if (STATS && sc.ltexs[sc.background_tex].kind == HRT_TEX_ENV) pc.env_lookups++;
        ps.result += ps.atten * background_value(sc, ps.d);
        return true;
    }
    if (STATS && sc.lprims[wh.prim].kind == HRT_PRIM_MESH) pc.mesh_hits++;
    DRec rec;
    hit_record(sc, wh, ps.o, ps.d, pr.quirks, pr.t_min, rec);
    vec3 emitted, attenuation, so, sd;
    const bool b = material_scatter(sc, rec, ps.d, ctx, emitted, attenuation, so, sd);
    ps.result += ps.atten * emitted;
    if (!b) return true;
    ps.atten *= attenuation;
    ps.o = so; ps.d = sd;
    ps.bounce++;
    return ps.bounce >= pr.max_depth;
}

// One iteration of the loop at main.cpp:43-76.  Returns true when the path has ended.
template <bool STATS>
__device__ inline bool path_segment(const DScene& sc, const hrt_params& pr, rng_ctx& ctx, PathState& ps, int* stack,
                                    PathCounters& pc) {
    ctx.bounce = (uint32_t)ps.bounce;
    pc.rays++;
    const WorldHit wh = world_hit<STATS>(sc, ps.o, ps.d, pr.t_min, __builtin_huge_valf(), pr.quirks, ctx, stack, pc.bvh);
    return path_shade<STATS>(sc, pr, ctx, ps, wh, pc);
}

// film.cpp:32-52 + 25-30
__device__ inline void film_resolve(vec3 c, uint8_t* out) {
    if (c.x != c.x) c.x = 0.0f;
    if (c.y != c.y) c.y = 0.0f;
    if (c.z != c.z) c.z = 0.0f;
    const float a = 2.51f, b = 0.03f, cc = 2.43f, dd = 0.59f, e = 0.14f;
    vec3 num = c * (a * c + b);
    vec3 den = c * (cc * c + dd) + e;
    vec3 q = num / den;
    q = vec3(gclamp(q.x, 0.0f, 1.0f), gclamp(q.y, 0.0f, 1.0f), gclamp(q.z, 0.0f, 1.0f));
    q = vec3(sqrtf(q.x), sqrtf(q.y), sqrtf(q.z));
    out[0] = static_cast<uint8_t>(256 * gclamp(q.x, 0.0f, 0.9999f));
    out[1] = static_cast<uint8_t>(256 * gclamp(q.y, 0.0f, 0.9999f));
    out[2] = static_cast<uint8_t>(256 * gclamp(q.z, 0.0f, 0.9999f));
}

}  // namespace hrt
