// oracle.cpp — CPU restatement of the reference's render loop.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing in the product path
// (hobbyraytracer_amd/, include/) links, loads or calls this file.  It is
// used by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline`
// leg as the checker / the timed CPU baseline, never as a fallback.
//
// PARITY PINNING: the reference binary cannot be built here (glm, assimp,
// yaml-cpp are empty submodules) and it ships no tests, so the
// only golden vectors are the six sphere-UV vectors of sphere.cpp:9-11, the
// ACES constants of film.cpp:40-46 and the quantisation rule of
// film.cpp:27-29.  Those are checked in tests/test_oracle_kats.py.  Every
// other result of this oracle is "parity unpinned": it follows the cited
// reference lines, but there is no reference output to compare it with.
//
// Structure mirrors the reference one class per class: virtual
// Hittable::hit, pointer-tree BVHNode built with the reference's
// random-axis / sort / median split (bvh.cpp:6-61), shared_ptr-free but
// otherwise the same call graph.  Arithmetic comes from csrc/hrt_glm.h (the
// glm restatement) and csrc/hrt_rng.h (counter RNG), compiled with
// -ffp-contract=off so the HIP kernels can be compared bit for bit.
//
// Input is the same `hrt_flat_scene` the HIP library consumes; this file
// rebuilds the reference's object graph from it (wrappers as nested
// Translate/Scale/RotateQuat/RotateY objects, meshes as ITriangle soups under
// a BVHNode tree) and ignores the flattened BVH nodes.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "../hobbyraytracer_amd/csrc/hrt_rng.h"
#include "../include/hrt.h"

using namespace hrt;

namespace {

// ---------------------------------------------------------------- context
struct Counters {
    uint64_t rays = 0, samples = 0, box_tests = 0, tri_tests = 0, mesh_hits = 0, env_lookups = 0;
};
thread_local rng_ctx g_ctx;
thread_local Counters g_cnt;
thread_local uint32_t g_quirks = HRT_QUIRKS_REFERENCE;

// ---------------------------------------------------------------- ray.h:3-13
struct ray {
    vec3 o, dir;
    ray() {}
    ray(vec3 origin, vec3 direction) : o(origin), dir(direction) {}
    vec3 at(float t) const { return o + (t * dir); }
};

// ---------------------------------------------------------------- aabb.h:7-60
struct AABB {
    vec3 mn, mx;
    AABB() : mn(0.0f), mx(0.0f) {}
    AABB(vec3 a, vec3 b) : mn(a), mx(b) {}
    // aabb.h:26-39
    bool hit(const ray& r, float t_min, float t_max) const {
        for (int a = 0; a < 3; a++) {
            float t0 = std::min((mn[a] - r.o[a]) / r.dir[a], (mx[a] - r.o[a]) / r.dir[a]);
            float t1 = std::max((mn[a] - r.o[a]) / r.dir[a], (mx[a] - r.o[a]) / r.dir[a]);
            t_min = std::max(t0, t_min);
            t_max = std::min(t1, t_max);
            if (t_max <= t_min) return false;
        }
        return true;
    }
    // aabb.h:41-56
    static AABB surroundingBox(const AABB& a, const AABB& b) {
        vec3 small(gmin(a.mn.x, b.mn.x), gmin(a.mn.y, b.mn.y), gmin(a.mn.z, b.mn.z));
        vec3 big(gmax(a.mx.x, b.mx.x), gmax(a.mx.y, b.mx.y), gmax(a.mx.z, b.mx.z));
        return AABB(small, big);
    }
};

struct Material;

// ---------------------------------------------------------------- hittable.h:8-32
struct hitRecord {
    vec3 p;
    vec3 normal;
    const Material* matPtr = nullptr;
    float t = 0.0f;
    float u = 0.0f, v = 0.0f;
    // The reference leaves this uninitialised (main.cpp:44, hittableList.cpp:6) and ITriangle::hit never sets it (Q-3): a mesh
    // hit carries the flag of the previous successful object of the list walk, and the initial value -- indeterminate in the
    // reference, defined `true` here -- when there was none.
    bool frontFace = true;
    int tri = -1;  // bookkeeping for the parity tests, not in the reference
    void setFaceNormal(const ray& r, vec3 outward_normal) {
        frontFace = dot(r.dir, outward_normal) < 0;
        normal = frontFace ? outward_normal : -outward_normal;
    }
};

struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const = 0;
    virtual bool boundingBox(AABB& out) = 0;
};

// ---------------------------------------------------------------- texture.h / texture.cpp
struct Texture {
    virtual ~Texture() {}
    virtual vec3 colourValue(float u, float v, vec3 p) const = 0;
    virtual bool isEnv() const { return false; }
};
struct SolidColourTexture : Texture {  // texture.h:18-21
    vec3 c;
    explicit SolidColourTexture(vec3 col) : c(col) {}
    vec3 colourValue(float, float, vec3) const override { return c; }
};
struct CheckeredTexture : Texture {  // texture.cpp:17-28
    const Texture *e, *o;
    CheckeredTexture(const Texture* even, const Texture* odd) : e(even), o(odd) {}
    vec3 colourValue(float u, float v, vec3 p) const override {
        float sines = gsin_wide(10 * p.x) * gsin_wide(10 * p.y) * gsin_wide(10 * p.z);
        if (sines < 0) return o->colourValue(u, v, p);
        return e->colourValue(u, v, p);
    }
};
struct ImageTexture : Texture {  // texture.cpp:53-74
    const uint8_t* data; int width, height;
    ImageTexture(const uint8_t* d, int w, int h) : data(d), width(w), height(h) {}
    vec3 colourValue(float u, float v, vec3) const override {
        if (width == 0 || height == 0) return vec3(0, 1, 1);
        u = gclamp(u, 0.0f, 1.0f);
        v = 1.0f - gclamp(v, 0.0f, 1.0f);
        int i = texel_index(u * width);
        int j = texel_index(v * height);
        if (i >= width) i = width - 1;
        if (j >= height) j = height - 1;
        const float colourScale = 1.0f / 255.0f;
        int pixel = j * (3 * width) + i * 3;
        return vec3(colourScale * data[pixel], colourScale * data[pixel + 1], colourScale * data[pixel + 2]);
    }
};
struct EnvironmentMap : Texture {  // texture.cpp:76-97
    const float* data; int width, height, channels;
    EnvironmentMap(const float* d, int w, int h, int c) : data(d), width(w), height(h), channels(c) {}
    bool isEnv() const override { return true; }
    vec3 colourValue(float u, float v, vec3) const override {
        if (width == 0 || height == 0) return vec3(0, 1, 1);
        u = gclamp(u, 0.0f, 1.0f);
        v = gclamp(v, 0.0f, 1.0f);
        int i = texel_index((u * (width - 1)) + 0.5f);
        int j = texel_index((v * (height - 1)) + 0.5f);
        size_t b = ((size_t)j * width + i) * channels;
        return vec3(data[b], data[b + 1], data[b + 2]);
    }
};

// ---------------------------------------------------------------- material.h
struct MatVec3 {  // material.h:10-35
    const Texture* tex = nullptr; vec3 c;
    vec3 valueAt(float u, float v, vec3 p) const { return tex ? tex->colourValue(u, v, p) : c; }
};
struct MatScalar {  // material.h:37-58
    const Texture* tex = nullptr; float c = 0.0f;
    float valueAt(float u, float v, vec3 p) const { return tex ? length(tex->colourValue(u, v, p)) : c; }
};

struct Material {
    virtual ~Material() {}
    virtual bool scatter(const ray& r_in, const hitRecord& rec, vec3& attenuation, ray& scattered) const = 0;
    virtual vec3 emitted(float, float, vec3) const { return vec3(0, 0, 0); }
};

// One Philox draw per scatter event: x = theta, y = z-uniform of sphericalRand,
// (z,w) = the 64-bit uniform of Dielectric's `linearRand(0.0, 1.0)`.
static u32x4 scatter_draw() { return rng_draw(g_ctx, RNG_SCATTER, 0); }

struct Isotropic : Material {  // material.h:73-89
    MatVec3 albedo;
    bool scatter(const ray&, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        scattered = ray(rec.p, ball_rand(g_ctx));
        attenuation = albedo.valueAt(rec.u, rec.v, rec.p);
        return true;
    }
};
struct DiffuseLight : Material {  // material.h:91-109
    MatVec3 emit; MatScalar s;
    bool scatter(const ray&, const hitRecord&, vec3&, ray&) const override { return false; }
    vec3 emitted(float u, float v, vec3 p) const override { return emit.valueAt(u, v, p) * s.valueAt(u, v, p); }
};
struct UVTest : Material {  // material.h:111-130
    bool scatter(const ray&, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        u32x4 d = scatter_draw();
        vec3 scatterDirection = rec.normal + spherical_rand(d.x, d.y);
        if (near_zero(scatterDirection)) scatterDirection = rec.normal;
        scattered = ray(rec.p, scatterDirection);
        attenuation = rec.normal;
        return true;
    }
};
struct Lambertian : Material {  // material.h:132-157
    MatVec3 albedo;
    bool scatter(const ray&, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        u32x4 d = scatter_draw();
        vec3 scatterDirection = rec.normal + spherical_rand(d.x, d.y);
        if (near_zero(scatterDirection)) scatterDirection = rec.normal;
        scattered = ray(rec.p, scatterDirection);
        attenuation = albedo.valueAt(rec.u, rec.v, rec.p);
        return true;
    }
};
struct Metal : Material {  // material.h:159-182
    MatVec3 albedo; MatScalar r;
    bool scatter(const ray& r_in, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        u32x4 d = scatter_draw();
        vec3 reflected = reflect(normalize(r_in.dir), normalize(rec.normal));
        float roughness = fabsf(r.valueAt(rec.u, rec.v, rec.p));  // glm::length(float) = abs
        roughness = roughness < 1 ? roughness : 1;
        scattered = ray(rec.p, reflected + roughness * spherical_rand(d.x, d.y) +
                                   vec3(std::numeric_limits<float>::epsilon()));
        attenuation = albedo.valueAt(rec.u, rec.v, rec.p);
        return dot(scattered.dir, normalize(rec.normal)) > 0;
    }
};
struct Dielectric : Material {  // material.h:199-242
    MatScalar ir, r;
    static double reflectance(double cosine, float refIdx) {  // material.h:236-241
        double r0 = (1 - refIdx) / (1 + refIdx);
        r0 = r0 * r0;
        return r0 + (1 - r0) * pow5(1 - cosine);
    }
    bool scatter(const ray& r_in, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        u32x4 d = scatter_draw();
        attenuation = vec3(1, 1, 1);
        float refractionRatio = rec.frontFace ? (1.0f / ir.valueAt(rec.u, rec.v, rec.p)) : ir.valueAt(rec.u, rec.v, rec.p);
        vec3 unitDirection = normalize(r_in.dir);
        double cosTheta = gmin(dot(-unitDirection, rec.normal), 1.0f);
        double sinTheta = sqrt(1.0 - cosTheta * cosTheta);
        bool cannot_refract = refractionRatio * sinTheta > 1.0;
        vec3 direction;
        double ref = reflectance(cosTheta, refractionRatio);
        if (cannot_refract || ref > u01d(d.z, d.w)) direction = reflect(unitDirection, rec.normal);
        else direction = refract(unitDirection, rec.normal, refractionRatio);
        scattered = ray(rec.p, direction + r.valueAt(rec.u, rec.v, rec.p) * spherical_rand(d.x, d.y));
        return true;
    }
};
struct PBR : Material {  // material.cpp:4-28
    Metal metal; Lambertian diffuse; const Texture* mix = nullptr;
    bool scatter(const ray& r_in, const hitRecord& rec, vec3& attenuation, ray& scattered) const override {
        bool m = length(mix->colourValue(rec.u, rec.v, rec.p)) > 0.5f;
        if (m) return metal.scatter(r_in, rec, attenuation, scattered);
        return diffuse.scatter(r_in, rec, attenuation, scattered);
    }
};

// ---------------------------------------------------------------- hittableList.cpp:4-37
struct HittableList : Hittable {
    std::vector<Hittable*> objects;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        hitRecord tempRec;
        bool hitAnything = false;
        float closest = t_max;
        for (const Hittable* object : objects) {
            if (object->hit(r, t_min, closest, tempRec)) {
                hitAnything = true;
                closest = tempRec.t;
                rec = tempRec;
            }
        }
        return hitAnything;
    }
    bool boundingBox(AABB& out) override {
        if (objects.empty()) return false;
        AABB tempBox; bool firstBox = true;
        for (Hittable* object : objects) {
            if (!object->boundingBox(tempBox)) return false;
            out = firstBox ? tempBox : AABB::surroundingBox(out, tempBox);
            firstBox = false;
        }
        return true;
    }
};

// ---------------------------------------------------------------- sphere.cpp
static void getSphereUV(vec3 p, float& u, float& v) {  // sphere.cpp:4-18
    const float pi = 3.14159265358979323846264338327950288f;
    float theta = gacos(-p.y);
    float phi = gatan2(-p.z, p.x) + pi;
    u = phi / (2 * pi);
    v = theta / pi;
}
struct Sphere : Hittable {
    vec3 center; float radius; const Material* matPtr;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {  // sphere.cpp:20-49
        vec3 oc = r.o - center;
        float a = length(r.dir) * length(r.dir);
        float half_b = dot(oc, r.dir);
        float c = length(oc) * length(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0) return false;
        float sqrtd = sqrtf(discriminant);
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || root > t_max) return false;
        }
        rec.t = root;
        rec.p = r.at(rec.t);
        vec3 outwardNormal = (rec.p - center) / radius;
        rec.setFaceNormal(r, outwardNormal);
        getSphereUV(outwardNormal, rec.u, rec.v);
        rec.matPtr = matPtr;
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override { out = AABB(center - vec3(radius), center + vec3(radius)); return true; }
};

// ---------------------------------------------------------------- triangle.cpp:4-55 (Triangle; meshes use ITriangle below)
struct Triangle : Hittable {
    vec3 v0, v1, v2; const Material* matPtr;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {  // triangle.cpp:4-40
        vec3 v0v1 = v1 - v0;
        vec3 v0v2 = v2 - v0;
        vec3 pV = normalize(cross(r.dir, v0v2));
        float d = dot(normalize(v0v1), pV);
        if (d < 0.0001f) return false;
        if (fabsf(d) < 0.0001f) return false;
        float invD = 1.0f / d;
        vec3 tV = normalize(r.o - v0);
        rec.u = dot(tV, pV) * invD;
        if (rec.u < 0 || rec.u > 1) return false;
        vec3 qV = cross(tV, normalize(v0v1));
        rec.v = dot(normalize(r.dir), qV) * invD;
        if (rec.v < 0 || rec.u + rec.v > 1) return false;
        rec.t = dot(v0v2, qV) * invD;
        if (rec.t < t_min) return false;
        if (rec.t > t_max) return false;
        rec.p = r.at(rec.t);
        rec.matPtr = matPtr;
        rec.setFaceNormal(r, cross(v0v1, v0v2));
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override {  // triangle.cpp:42-55
        vec3 mn(gmin(gmin(v0.x, v1.x), v2.x), gmin(gmin(v0.y, v1.y), v2.y), gmin(gmin(v0.z, v1.z), v2.z));
        vec3 mx(gmax(gmax(v0.x, v1.x), v2.x), gmax(gmax(v0.y, v1.y), v2.y), gmax(gmax(v0.z, v1.z), v2.z));
        out = AABB(mn - vec3(0.0001f), mx + vec3(0.0001f));
        return true;
    }
};

// ---------------------------------------------------------------- aarect.h
struct YZRect : Hittable {  // aarect.h:12-39
    float y0, y1, z0, z1, k; const Material* mp;
    YZRect(float a, float b, float c, float d, float kk, const Material* m) : y0(a), y1(b), z0(c), z1(d), k(kk), mp(m) {}
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        float t = (k - r.o.x) / r.dir.x;
        if (t < t_min || t > t_max) return false;
        float y = r.o.y + t * r.dir.y;
        float z = r.o.z + t * r.dir.z;
        if (y < y0 || y > y1 || z < z0 || z > z1) return false;
        rec.u = (y - y0) / (y1 - y0);
        rec.v = (z - z0) / (z1 - z0);
        rec.t = t;
        rec.setFaceNormal(r, vec3(1, 0, 0));
        rec.matPtr = mp;
        rec.p = r.at(t);
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override { out = AABB(vec3(k - 0.0001f, y0, z0), vec3(k + 0.0001f, y1, z1)); return true; }
};
struct XZRect : Hittable {  // aarect.h:59-86
    float x0, x1, z0, z1, k; const Material* mp;
    XZRect(float a, float b, float c, float d, float kk, const Material* m) : x0(a), x1(b), z0(c), z1(d), k(kk), mp(m) {}
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        float t = (k - r.o.y) / r.dir.y;
        if (t < t_min || t > t_max) return false;
        float x = r.o.x + t * r.dir.x;
        float z = r.o.z + t * r.dir.z;
        if (x < x0 || x > x1 || z < z0 || z > z1) return false;
        rec.u = (x - x0) / (x1 - x0);
        rec.v = (z - z0) / (z1 - z0);
        rec.t = t;
        rec.setFaceNormal(r, vec3(0, 1, 0));
        rec.matPtr = mp;
        rec.p = r.at(t);
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override { out = AABB(vec3(x0, k - 0.0001f, z0), vec3(x1, k + 0.0001f, z1)); return true; }
};
struct XYRect : Hittable {  // aarect.h:106-133
    float x0, x1, y0, y1, k; const Material* mp;
    XYRect(float a, float b, float c, float d, float kk, const Material* m) : x0(a), x1(b), y0(c), y1(d), k(kk), mp(m) {}
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        float t = (k - r.o.z) / r.dir.z;
        if (t < t_min || t > t_max) return false;
        float x = r.o.x + t * r.dir.x;
        float y = r.o.y + t * r.dir.y;
        if (x < x0 || x > x1 || y < y0 || y > y1) return false;
        rec.u = (x - x0) / (x1 - x0);
        rec.v = (y - y0) / (y1 - y0);
        rec.t = t;
        rec.setFaceNormal(r, vec3(0, 0, 1));
        rec.matPtr = mp;
        rec.p = r.at(t);
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override { out = AABB(vec3(x0, y0, k - 0.0001f), vec3(x1, y1, k + 0.0001f)); return true; }
};

// ---------------------------------------------------------------- box.h:27-55
struct Box : Hittable {
    vec3 boxMin, boxMax; HittableList sides; std::vector<std::unique_ptr<Hittable>> own;
    Box(vec3 mn, vec3 mx, const Material* m) : boxMin(mn), boxMax(mx) {
        auto add = [&](Hittable* h) { own.emplace_back(h); sides.objects.push_back(h); };
        add(new XYRect(mn.x, mx.x, mn.y, mx.y, mx.z, m));
        add(new XYRect(mn.x, mx.x, mn.y, mx.y, mn.z, m));
        add(new XZRect(mn.x, mx.x, mn.z, mx.z, mx.y, m));
        add(new XZRect(mn.x, mx.x, mn.z, mx.z, mn.y, m));
        add(new YZRect(mn.y, mx.y, mn.z, mx.z, mx.x, m));
        add(new YZRect(mn.y, mx.y, mn.z, mx.z, mn.x, m));
    }
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override { return sides.hit(r, t_min, t_max, rec); }
    bool boundingBox(AABB& out) override { out = AABB(boxMin, boxMax); return true; }
};

// ---------------------------------------------------------------- triangle.cpp:57-151
struct ITriangle : Hittable {
    vec3 vertices[3], normals[3]; vec2 uvs[3]; const Material* matPtr; AABB bBox; int index;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        g_cnt.tri_tests++;
        vec3 d = r.dir;
        vec3 o = r.o;
        vec3 p0t = vertices[0] - o;
        vec3 p1t = vertices[1] - o;
        vec3 p2t = vertices[2] - o;
        int kZ;
        if (g_quirks & HRT_Q4_SHEAR_FROM_ORIGIN) {
            kZ = o.x > o.z ? (o.x > o.y ? 0 : 1) : 2;  // triangle.cpp:70
        } else {  // PBRT: MaxDimension(Abs(d))
            float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
            kZ = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
        }
        int kX = kZ + 1 == 3 ? 0 : kZ + 1;
        int kY = kX + 1 == 3 ? 0 : kX + 1;
        d = vec3(d[kX], d[kY], d[kZ]);
        p0t = vec3(p0t[kX], p0t[kY], p0t[kZ]);
        p1t = vec3(p1t[kX], p1t[kY], p1t[kZ]);
        p2t = vec3(p2t[kX], p2t[kY], p2t[kZ]);
        float sX = -d.x / d.z;
        float sY = -d.y / d.z;
        float sZ = 1.0f / d.z;
        p0t.x += sX * p0t.z; p0t.y += sY * p0t.z;
        p1t.x += sX * p1t.z; p1t.y += sY * p1t.z;
        p2t.x += sX * p2t.z; p2t.y += sY * p2t.z;
        float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
        float det = e0 + e1 + e2;
        if (det == 0) return false;
        p0t.z *= sZ; p1t.z *= sZ; p2t.z *= sZ;
        float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0 && (tScaled >= 0 || tScaled < t_max * det)) return false;
        else if (det > 0 && (tScaled <= 0 || tScaled > t_max * det)) return false;
        float invDet = 1 / det;
        float b0 = e0 * invDet;
        float b1 = e1 * invDet;
        float b2 = e2 * invDet;
        float t = tScaled * invDet;
        if (!(g_quirks & HRT_Q2_TRI_NO_TMIN) && t < t_min) return false;  // Q-2 fixed
        rec.t = t;
        rec.p = r.at(rec.t);
        rec.matPtr = matPtr;
        vec3 normal = b0 * normals[0] + b1 * normals[1] + b2 * normals[2];
        vec2 uv = b0 * uvs[0] + b1 * uvs[1] + b2 * uvs[2];
        rec.normal = normal;
        rec.u = uv.x;
        rec.v = uv.y;
        rec.tri = index;
        // Q-3: triangle.cpp:118-128 never write rec.frontFace.  `rec` is HittableList::hit's tempRec (hittableList.cpp:6-16, handed
        // down through Mesh::hit and BVHNode::hit), shared by all objects of the walk: the flag keeps what the previous successful
        // object wrote (a wrapper around the mesh overwrites it afterwards, translate.cpp:16), or its initial value.
        if (!(g_quirks & HRT_Q3_TRI_NO_FACE)) rec.setFaceNormal(r, normal);   // Q-3 fixed
        return true;
    }
    bool boundingBox(AABB& out) override {  // triangle.cpp:133-151
        float minX = gmin(gmin(vertices[0].x, vertices[1].x), vertices[2].x);
        float minY = gmin(gmin(vertices[0].y, vertices[1].y), vertices[2].y);
        float minZ = gmin(gmin(vertices[0].z, vertices[1].z), vertices[2].z);
        float maxX = gmax(gmax(vertices[0].x, vertices[1].x), vertices[2].x);
        float maxY = gmax(gmax(vertices[0].y, vertices[1].y), vertices[2].y);
        float maxZ = gmax(gmax(vertices[0].z, vertices[1].z), vertices[2].z);
        bBox = AABB(vec3(minX - 0.0001f, minY - 0.0001f, minZ - 0.0001f), vec3(maxX + 0.0001f, maxY + 0.0001f, maxZ + 0.0001f));
        out = bBox;
        return true;
    }
};

// ---------------------------------------------------------------- bvh.cpp:6-90
struct BVHNode : Hittable {
    Hittable* left = nullptr; Hittable* right = nullptr; AABB box;
    std::unique_ptr<BVHNode> ownL, ownR;
    static bool boxCompare(Hittable* a, Hittable* b, int axis) {
        AABB boxA, boxB;
        a->boundingBox(boxA); b->boundingBox(boxB);
        return boxA.mn[axis] < boxB.mn[axis];
    }
    BVHNode(std::vector<Hittable*>& src, size_t start, size_t end, uint32_t& serial) {
        // bvh.cpp:10  int a = glm::linearRand(0, 2)  ->  u32 % 3 (setup-time draw)
        u32x4 u = philox4x32_10(serial++, 0, 0, RNG_BUILD, 0, 0);
        int a = (int)(u.x % 3u);
        auto comparator = [a](Hittable* x, Hittable* y) { return boxCompare(x, y, a); };
        size_t n = end - start;
        if (n == 1) {
            left = right = src[start];
        } else if (n == 2) {
            if (comparator(src[start], src[start + 1])) { left = src[start]; right = src[start + 1]; }
            else { left = src[start + 1]; right = src[start]; }
        } else if (n != 0) {
            std::sort(src.begin() + start, src.begin() + end, comparator);
            size_t mid = start + n / 2;
            ownL.reset(new BVHNode(src, start, mid, serial));
            ownR.reset(new BVHNode(src, mid, end, serial));
            left = ownL.get(); right = ownR.get();
        } else {
            return;
        }
        AABB boxLeft, boxRight;
        left->boundingBox(boxLeft);
        right->boundingBox(boxRight);
        box = AABB::surroundingBox(boxLeft, boxRight);
    }
    bool boundingBox(AABB& out) override { out = box; return true; }
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {  // bvh.cpp:69-78
        g_cnt.box_tests++;
        if (!box.hit(r, t_min, t_max)) return false;
        if (!left) return false;  // empty mesh (mesh.cpp:21,38 / bvh.cpp:45-48): degenerate box already rejects
        bool hitLeft = left->hit(r, t_min, t_max, rec);
        bool hitRight = right->hit(r, t_min, hitLeft ? rec.t : t_max, rec);
        return hitLeft || hitRight;
    }
};

// ---------------------------------------------------------------- mesh.cpp:13-51
struct Mesh : Hittable {
    std::vector<ITriangle> tris; std::unique_ptr<BVHNode> tree;
    Mesh(const float* pos, const float* nrm, const float* uv, uint32_t count, const Material* m) {
        tris.resize(count);
        for (uint32_t i = 0; i < count; ++i) {
            ITriangle& t = tris[i];
            for (int k = 0; k < 3; ++k) {
                t.vertices[k] = vec3(pos[9 * i + 3 * k], pos[9 * i + 3 * k + 1], pos[9 * i + 3 * k + 2]);
                t.normals[k] = vec3(nrm[9 * i + 3 * k], nrm[9 * i + 3 * k + 1], nrm[9 * i + 3 * k + 2]);
                t.uvs[k] = vec2(uv[6 * i + 2 * k], uv[6 * i + 2 * k + 1]);
            }
            t.matPtr = m; t.index = (int)i;
        }
        std::vector<Hittable*> strip;
        for (auto& t : tris) strip.push_back(&t);
        uint32_t serial = 0;
        tree.reset(new BVHNode(strip, 0, strip.size(), serial));
    }
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override { return tree->hit(r, t_min, t_max, rec); }
    bool boundingBox(AABB& out) override { return tree->boundingBox(out); }
};

// ---------------------------------------------------------------- translate.cpp:7-19
struct Translate : Hittable {
    Hittable* ptr; vec3 offset;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        ray movedR(r.o - offset, r.dir);
        if (!ptr->hit(movedR, t_min, t_max, rec)) return false;
        rec.p += offset;
        rec.setFaceNormal(movedR, rec.normal);
        return true;
    }
    bool boundingBox(AABB& out) override {
        if (!ptr->boundingBox(out)) return false;
        out = AABB(out.mn + offset, out.mx + offset);
        return true;
    }
};
// ---------------------------------------------------------------- scale.cpp:11-27
struct Scale : Hittable {
    Hittable* ptr; vec3 factor;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        ray scaledRay(r.o / factor, r.dir / factor);
        if (!ptr->hit(scaledRay, t_min, t_max, rec)) return false;
        rec.p *= factor;
        rec.setFaceNormal(scaledRay, rec.normal);
        return true;
    }
    bool boundingBox(AABB& out) override { return ptr->boundingBox(out); }  // scale.cpp:29-33 (unscaled, Q-7)
};
// ---------------------------------------------------------------- rotateQuat.cpp:44-66
struct RotateQuat : Hittable {
    Hittable* ptr; quat rotation;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        quat invRotation = conjugate(rotation);
        vec3 newOrigin = rotate(invRotation, r.o);
        vec3 newDirection = rotate(invRotation, r.dir);
        if (g_quirks & HRT_Q1_ROTQ_NORMALIZE) newDirection = normalize(newDirection);  // Q-1
        ray rotatedRay(newOrigin, newDirection);
        if (!ptr->hit(rotatedRay, t_min, t_max, rec)) return false;
        rec.p = rotate(rotation, rec.p);
        rec.normal = rotate(rotation, rec.normal);
        rec.setFaceNormal(rotatedRay, rec.normal);
        return true;
    }
    bool boundingBox(AABB& out) override { return ptr->boundingBox(out); }  // box unused by the world list
};
// ---------------------------------------------------------------- rotateY.cpp:44-75
struct RotateY : Hittable {
    Hittable* ptr; float sinTheta, cosTheta;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        vec3 origin = r.o, direction = r.dir;
        origin.x = cosTheta * r.o.x - sinTheta * r.o.z;
        origin.z = sinTheta * r.o.x + cosTheta * r.o.z;
        direction.x = cosTheta * r.dir.x - sinTheta * r.dir.z;
        direction.z = sinTheta * r.dir.x + cosTheta * r.dir.z;
        ray rotatedR(origin, direction);
        if (!ptr->hit(rotatedR, t_min, t_max, rec)) return false;
        vec3 p = rec.p, normal = rec.normal;
        p.x = cosTheta * rec.p.x + sinTheta * rec.p.z;
        p.z = -sinTheta * rec.p.x + cosTheta * rec.p.z;
        normal.x = cosTheta * rec.normal.x + sinTheta * rec.normal.z;
        normal.z = -sinTheta * rec.normal.x + cosTheta * rec.normal.z;
        rec.p = p;
        rec.setFaceNormal(rotatedR, normal);
        return true;
    }
    bool boundingBox(AABB& out) override { return ptr->boundingBox(out); }
};

// ---------------------------------------------------------------- constantMedium.cpp:4-38
struct ConstantMedium : Hittable {
    Hittable* boundary; const Material* phaseFunction; float negInvDensity; uint32_t prim_index;
    bool hit(const ray& r, float t_min, float t_max, hitRecord& rec) const override {
        hitRecord rec1, rec2;
        const float INF = std::numeric_limits<float>::infinity();
        if (!boundary->hit(r, -INF, INF, rec1)) return false;
        if (!boundary->hit(r, rec1.t + 0.0001f, INF, rec2)) return false;
        if (rec1.t < t_min) rec1.t = t_min;
        if (rec2.t > t_max) rec2.t = t_max;
        if (rec1.t >= rec2.t) return false;
        if (rec1.t < 0) rec1.t = 0;
        const float ray_length = length(r.dir);
        const float distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        u32x4 u = rng_draw(g_ctx, RNG_MEDIUM, prim_index);
        const float hit_distance = negInvDensity * glog(linear_rand(u.x, 0.0f, 1.0f));
        if (hit_distance > distance_inside_boundary) return false;
        rec.t = rec1.t + hit_distance / ray_length;
        rec.p = r.at(rec.t);
        rec.normal = vec3(1, 0, 0);
        rec.frontFace = true;
        rec.matPtr = phaseFunction;
        // constantMedium.cpp:30-36 leaves rec.u / rec.v untouched: in the reference they hold whatever the
        // previous successful object of the world list wrote into HittableList::hit's tempRec (or garbage).
        // Defined here as 0 (only a texture-valued Isotropic albedo could ever read them).
        rec.u = 0.0f; rec.v = 0.0f;
        rec.tri = -1;
        return true;
    }
    bool boundingBox(AABB& out) override { return boundary->boundingBox(out); }
};

// ---------------------------------------------------------------- object graph from the flat scene
struct World {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Hittable>> own;
    HittableList list;                 // scene.cpp:376-379: the world IS a HittableList
    std::vector<Hittable*> top;        // top[i] = outermost object of prim i
    const Texture* background = nullptr;
};

static MatVec3 mk_mv3(const World& w, const hrt_matvec3& m) {
    MatVec3 r; r.c = vec3(m.c[0], m.c[1], m.c[2]); r.tex = m.tex >= 0 ? w.textures[m.tex].get() : nullptr; return r;
}
static MatScalar mk_ms(const World& w, const hrt_matscalar& m) {
    MatScalar r; r.c = m.c; r.tex = m.tex >= 0 ? w.textures[m.tex].get() : nullptr; return r;
}

static Hittable* build_leaf(World& w, const hrt_flat_scene* fs, int kind, const float* p, int mesh, const Material* m) {
    Hittable* h = nullptr;
    switch (kind) {
        case HRT_PRIM_SPHERE: { auto* s = new Sphere; s->center = vec3(p[0], p[1], p[2]); s->radius = p[3]; s->matPtr = m; h = s; break; }
        case HRT_PRIM_XY_RECT: h = new XYRect(p[0], p[1], p[2], p[3], p[4], m); break;
        case HRT_PRIM_XZ_RECT: h = new XZRect(p[0], p[1], p[2], p[3], p[4], m); break;
        case HRT_PRIM_YZ_RECT: h = new YZRect(p[0], p[1], p[2], p[3], p[4], m); break;
        case HRT_PRIM_BOX: h = new Box(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]), m); break;
        case HRT_PRIM_TRIANGLE: { auto* t = new Triangle; t->v0 = vec3(p[0], p[1], p[2]); t->v1 = vec3(p[3], p[4], p[5]); t->v2 = vec3(p[6], p[7], p[8]); t->matPtr = m; h = t; break; }
        case HRT_PRIM_MESH: {
            const hrt_mesh& mm = fs->meshes[mesh];
            h = new Mesh(fs->tri_pos + 9ull * mm.tri_first, fs->tri_nrm + 9ull * mm.tri_first, fs->tri_uv + 6ull * mm.tri_first,
                         mm.tri_count, m);
            break;
        }
        default: break;
    }
    if (h) w.own.emplace_back(h);
    return h;
}

static std::unique_ptr<World> build_world(const hrt_flat_scene* fs) {
    std::unique_ptr<World> w(new World);
    // textures (two passes so checker children may reference later entries)
    w->textures.resize(fs->n_textures);
    for (uint32_t i = 0; i < fs->n_textures; ++i) {
        const hrt_texture& t = fs->textures[i];
        if (t.kind == HRT_TEX_SOLID) w->textures[i].reset(new SolidColourTexture(vec3(t.c[0], t.c[1], t.c[2])));
        else if (t.kind == HRT_TEX_IMAGE) w->textures[i].reset(new ImageTexture(fs->texels_u8 + t.offset, t.width, t.height));
        else if (t.kind == HRT_TEX_ENV) w->textures[i].reset(new EnvironmentMap(fs->texels_f32 + t.offset, t.width, t.height, t.channels));
    }
    for (uint32_t i = 0; i < fs->n_textures; ++i) {
        const hrt_texture& t = fs->textures[i];
        if (t.kind == HRT_TEX_CHECKER) w->textures[i].reset(new CheckeredTexture(w->textures[t.even].get(), w->textures[t.odd].get()));
    }
    w->background = w->textures[fs->background_tex].get();
    for (uint32_t i = 0; i < fs->n_materials; ++i) {
        const hrt_material& m = fs->materials[i];
        Material* mat = nullptr;
        switch (m.kind) {
            case HRT_MAT_LAMBERTIAN: { auto* x = new Lambertian; x->albedo = mk_mv3(*w, m.albedo); mat = x; break; }
            case HRT_MAT_METAL: { auto* x = new Metal; x->albedo = mk_mv3(*w, m.albedo); x->r = mk_ms(*w, m.s0); mat = x; break; }
            case HRT_MAT_DIELECTRIC: { auto* x = new Dielectric; x->ir = mk_ms(*w, m.s0); x->r = mk_ms(*w, m.s1); mat = x; break; }
            case HRT_MAT_DIFFUSE_LIGHT: { auto* x = new DiffuseLight; x->emit = mk_mv3(*w, m.albedo); x->s = mk_ms(*w, m.s0); mat = x; break; }
            case HRT_MAT_ISOTROPIC: { auto* x = new Isotropic; x->albedo = mk_mv3(*w, m.albedo); mat = x; break; }
            case HRT_MAT_PBR: {
                auto* x = new PBR;
                x->metal.albedo = mk_mv3(*w, m.albedo); x->metal.r = mk_ms(*w, m.s0);
                x->diffuse.albedo = mk_mv3(*w, m.albedo);
                x->mix = w->textures[m.mix_tex].get();
                mat = x; break;
            }
            default: mat = new UVTest; break;
        }
        w->materials.emplace_back(mat);
    }
    for (uint32_t i = 0; i < fs->n_prims; ++i) {
        const hrt_prim& pr = fs->prims[i];
        const Material* m = w->materials[pr.material].get();
        Hittable* h = nullptr;
        if (pr.kind == HRT_PRIM_MEDIUM) {
            Hittable* b = build_leaf(*w, fs, pr.boundary_kind, pr.p, -1, m);
            auto* cm = new ConstantMedium; cm->boundary = b; cm->phaseFunction = m; cm->negInvDensity = -1 / pr.density; cm->prim_index = i;
            w->own.emplace_back(cm); h = cm;
        } else {
            h = build_leaf(*w, fs, pr.kind, pr.p, pr.mesh, m);
        }
        // wrappers: xf[0] is outermost, so wrap from the innermost (last) outwards
        for (int k = pr.n_xforms - 1; k >= 0; --k) {
            const hrt_xform& x = pr.xf[k];
            Hittable* wr = nullptr;
            if (x.kind == HRT_XF_TRANSLATE) { auto* t = new Translate; t->ptr = h; t->offset = vec3(x.v[0], x.v[1], x.v[2]); wr = t; }
            else if (x.kind == HRT_XF_SCALE) { auto* t = new Scale; t->ptr = h; t->factor = vec3(x.v[0], x.v[1], x.v[2]); wr = t; }
            else if (x.kind == HRT_XF_ROTATE_QUAT) { auto* t = new RotateQuat; t->ptr = h; t->rotation.x = x.v[0]; t->rotation.y = x.v[1]; t->rotation.z = x.v[2]; t->rotation.w = x.v[3]; wr = t; }
            else { auto* t = new RotateY; t->ptr = h; t->sinTheta = x.v[0]; t->cosTheta = x.v[1]; wr = t; }
            w->own.emplace_back(wr); h = wr;
        }
        w->top.push_back(h);
        w->list.objects.push_back(h);
    }
    return w;
}

// world->hit with the prim index of the winner (same loop as HittableList::hit)
static bool world_hit(const World& w, const ray& r, float t_min, float t_max, hitRecord& rec, int& prim) {
    hitRecord tempRec;
    bool hitAnything = false;
    float closest = t_max;
    prim = -1;
    for (size_t i = 0; i < w.list.objects.size(); ++i) {
        if (w.list.objects[i]->hit(r, t_min, closest, tempRec)) {
            hitAnything = true; closest = tempRec.t; rec = tempRec; prim = (int)i;
        }
    }
    return hitAnything;
}

// ---------------------------------------------------------------- main.cpp:38-79
static vec3 rayColour(ray r, const World& w, const hrt_flat_scene* fs, int max_depth, float t_min) {
    vec3 currentAttenuation(1.0f);
    vec3 result(0.0f);
    const float pi = 3.14159265358979323846264338327950288f;
    for (int i = 0; i < max_depth; ++i) {
        g_ctx.bounce = (uint32_t)i;
        g_cnt.rays++;
        hitRecord rec; int prim;
        if (!world_hit(w, r, t_min, std::numeric_limits<float>::infinity(), rec, prim)) {
            vec3 nD = normalize(r.dir);
            float phi = gatan2(nD.z, nD.x);
            float theta = gacos(nD.y);
            float u = phi / (2 * pi) + 0.5f;
            float v = theta / pi;
            if (w.background->isEnv()) g_cnt.env_lookups++;
            result += currentAttenuation * w.background->colourValue(u, v, vec3(0.0f));
            break;
        }
        if (fs->prims[prim].kind == HRT_PRIM_MESH) g_cnt.mesh_hits++;
        ray scattered;
        vec3 attenuation;
        vec3 emitted = rec.matPtr->emitted(rec.u, rec.v, rec.p);
        bool b = rec.matPtr->scatter(r, rec, attenuation, scattered);
        if (!b) { result += currentAttenuation * emitted; break; }
        result += currentAttenuation * emitted;
        currentAttenuation *= attenuation;
        r = scattered;
    }
    return result;
}

// camera.h:29-39
static ray getRay(const hrt_camera* c, float s, float t, bool thin_lens) {
    vec3 origin(c->origin[0], c->origin[1], c->origin[2]);
    vec3 llc(c->lower_left[0], c->lower_left[1], c->lower_left[2]);
    vec3 hor(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    vec3 ver(c->vertical[0], c->vertical[1], c->vertical[2]);
    vec3 offset(0.0f);                     // camera.h:34: rd = {0, 0, 0}  // glm::circularRand(lensRadius)
    if (thin_lens) {                       // HRT_FLAG_THIN_LENS: that commented-out call, camera.h:34-35
        u32x4 l = rng_draw(g_ctx, RNG_LENS, 0);
        float rx, ry;
        circular_rand(l.x, c->lens_radius, rx, ry);
        offset = vec3(c->lens_u[0], c->lens_u[1], c->lens_u[2]) * rx + vec3(c->lens_v[0], c->lens_v[1], c->lens_v[2]) * ry;
    }
    return ray(origin + offset, llc + s * hor + t * ver - origin - offset);
}

// film.cpp:32-52
static vec3 tonemap(vec3 c) {
    if (c.x != c.x) c.x = 0.0f;
    if (c.y != c.y) c.y = 0.0f;
    if (c.z != c.z) c.z = 0.0f;
    float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
    vec3 num = c * (a * c + b);
    vec3 den = c * (cc * c + d) + e;
    vec3 q = num / den;
    q = vec3(gclamp(q.x, 0.0f, 1.0f), gclamp(q.y, 0.0f, 1.0f), gclamp(q.z, 0.0f, 1.0f));
    return vec3(sqrtf(q.x), sqrtf(q.y), sqrtf(q.z));
}
// film.cpp:25-30
static void writeColour(vec3 c, uint8_t* p) {
    p[0] = static_cast<uint8_t>(256 * gclamp(c.x, 0.0f, 0.9999f));
    p[1] = static_cast<uint8_t>(256 * gclamp(c.y, 0.0f, 0.9999f));
    p[2] = static_cast<uint8_t>(256 * gclamp(c.z, 0.0f, 0.9999f));
}

}  // namespace

// =============================================================== C entry points (tests / bench only)
extern "C" {

struct oracle_world { std::unique_ptr<World> w; const hrt_flat_scene* fs; };

// Builds the reference-style object graph (incl. the pointer-tree BVHs).  The
// flat scene's arrays must outlive the handle.
oracle_world* oracle_world_create(const hrt_flat_scene* fs) {
    oracle_world* h = new oracle_world;
    h->fs = fs;
    h->w = build_world(fs);
    return h;
}
void oracle_world_destroy(oracle_world* h) { delete h; }

// render() of main.cpp:81-140 for tile (x0,y0,w,h); writes the per-pixel mean
// linear radiance (the value handed to Film::tonemap at main.cpp:128).
int oracle_render_tile(oracle_world* h, const hrt_camera* cam, const hrt_params* pr, hrt_rect tile, float* out,
                       hrt_stats* stats, int n_threads) {
    if (!h || !cam || !pr || !out) return 1;
    if (n_threads < 1) n_threads = 1;
    const int W = pr->width, H = pr->height;
    std::atomic<int> next_row(0);
    std::vector<Counters> cnts(n_threads);
    auto worker = [&](int tid) {
        g_quirks = pr->quirks;
        g_cnt = Counters();
        g_ctx.seed_lo = pr->seed_lo; g_ctx.seed_hi = pr->seed_hi;
        for (;;) {
            int ry = next_row.fetch_add(1);
            if (ry >= tile.h) break;
            int row = tile.y0 + ry;
            for (int rx = 0; rx < tile.w; ++rx) {
                int px = tile.x0 + rx;
                int pIdx = row * W + px;
                vec3 pixelColour(0.0f);
                int x = pIdx % W;            // main.cpp:115
                int y = H - pIdx / W;        // main.cpp:116
                for (int s = 0; s < pr->samples; ++s) {
                    g_ctx.pixel = (uint32_t)pIdx; g_ctx.sample = (uint32_t)s; g_ctx.bounce = 0;
                    u32x4 j = rng_draw(g_ctx, RNG_JITTER, 0);
                    float u = ((float)x + linear_rand(j.x, 0.0f, 1.0f)) / (W - 1);
                    float v = ((float)y + linear_rand(j.y, 0.0f, 1.0f)) / (H - 1);
                    g_cnt.samples++;
                    pixelColour += rayColour(getRay(cam, u, v, (pr->flags & HRT_FLAG_THIN_LENS) != 0), *h->w, h->fs, pr->max_depth, pr->t_min);
                }
                pixelColour = pixelColour / static_cast<float>(pr->samples);
                float* o = out + 3 * ((size_t)ry * tile.w + rx);
                o[0] = pixelColour.x; o[1] = pixelColour.y; o[2] = pixelColour.z;
            }
        }
        cnts[tid] = g_cnt;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (stats) {
        hrt_stats s = {};
        for (auto& c : cnts) {
            s.rays += c.rays; s.samples += c.samples; s.box_tests += c.box_tests; s.tri_tests += c.tri_tests;
            s.mesh_hits += c.mesh_hits; s.env_lookups += c.env_lookups;
        }
        *stats = s;
    }
    return 0;
}

// world->hit(r, t_min, t_max, rec) of main.cpp:45 for n rays.
int oracle_closest_hit(oracle_world* h, const hrt_params* pr, int64_t n, const float* o, const float* d, float t_min,
                       float t_max, uint32_t pixel0, hrt_hit* out) {
    g_quirks = pr->quirks;
    g_ctx.seed_lo = pr->seed_lo; g_ctx.seed_hi = pr->seed_hi;
    for (int64_t i = 0; i < n; ++i) {
        g_ctx.pixel = pixel0 + (uint32_t)i; g_ctx.sample = 0; g_ctx.bounce = 0;
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        hitRecord rec; int prim;
        hrt_hit hh = {};
        if (world_hit(*h->w, r, t_min, t_max, rec, prim)) {
            hh.t = rec.t; hh.prim = prim; hh.tri = rec.tri; hh.front_face = rec.frontFace ? 1 : 0;
            hh.p[0] = rec.p.x; hh.p[1] = rec.p.y; hh.p[2] = rec.p.z;
            hh.normal[0] = rec.normal.x; hh.normal[1] = rec.normal.y; hh.normal[2] = rec.normal.z;
            hh.u = rec.u; hh.v = rec.v;
        } else {
            hh.prim = -1; hh.tri = -1;
        }
        out[i] = hh;
    }
    return 0;
}

// Material::scatter (material.h) at the first hit of n rays, for the statistical checks of the scatter distributions against
// float64 (tests/test_oracle_kats.py): ray i draws as (pixel0 + i, sample 0, bounce 0).  flag: -1 miss, 0 scatter() returned
// false (emitter, absorbed metal), 1 scattered.
int oracle_scatter(oracle_world* h, const hrt_params* pr, int64_t n, const float* o, const float* d, uint32_t pixel0, float* out_dir,
                   float* out_atten, int32_t* out_flag, hrt_hit* out_hit) {
    g_quirks = pr->quirks;
    g_ctx.seed_lo = pr->seed_lo; g_ctx.seed_hi = pr->seed_hi;
    for (int64_t i = 0; i < n; ++i) {
        g_ctx.pixel = pixel0 + (uint32_t)i; g_ctx.sample = 0; g_ctx.bounce = 0;
        ray r(vec3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), vec3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
        hitRecord rec; int prim;
        out_flag[i] = -1;
        for (int k = 0; k < 3; ++k) { out_dir[3 * i + k] = 0.0f; out_atten[3 * i + k] = 0.0f; }
        hrt_hit hh = {};
        hh.prim = -1; hh.tri = -1;
        if (world_hit(*h->w, r, pr->t_min, std::numeric_limits<float>::infinity(), rec, prim)) {
            hh.t = rec.t; hh.prim = prim; hh.tri = rec.tri; hh.front_face = rec.frontFace ? 1 : 0;
            hh.p[0] = rec.p.x; hh.p[1] = rec.p.y; hh.p[2] = rec.p.z;
            hh.normal[0] = rec.normal.x; hh.normal[1] = rec.normal.y; hh.normal[2] = rec.normal.z;
            ray scattered; vec3 attenuation(0, 0, 0);
            const bool ok = rec.matPtr->scatter(r, rec, attenuation, scattered);
            out_flag[i] = ok ? 1 : 0;
            if (ok) { out_dir[3 * i] = scattered.dir.x; out_dir[3 * i + 1] = scattered.dir.y; out_dir[3 * i + 2] = scattered.dir.z; }
            out_atten[3 * i] = attenuation.x; out_atten[3 * i + 1] = attenuation.y; out_atten[3 * i + 2] = attenuation.z;
        }
        if (out_hit) out_hit[i] = hh;
    }
    return 0;
}

// Debug aid for the parity tests: the segments of ONE path (pixel pIdx, sample s): per segment 6 floats
// (ray o, d) and 3 values (prim, tri, t).  Returns the number of segments written (<= max_seg).
int oracle_trace_path(oracle_world* h, const hrt_camera* cam, const hrt_params* pr, int pIdx, int s, int max_seg, float* rays, float* hits) {
    g_quirks = pr->quirks;
    g_ctx.seed_lo = pr->seed_lo; g_ctx.seed_hi = pr->seed_hi; g_ctx.pixel = (uint32_t)pIdx; g_ctx.sample = (uint32_t)s; g_ctx.bounce = 0;
    const int W = pr->width, H = pr->height;
    int x = pIdx % W, y = H - pIdx / W;
    u32x4 j = rng_draw(g_ctx, RNG_JITTER, 0);
    float u = ((float)x + linear_rand(j.x, 0.0f, 1.0f)) / (W - 1);
    float v = ((float)y + linear_rand(j.y, 0.0f, 1.0f)) / (H - 1);
    ray r = getRay(cam, u, v, (pr->flags & HRT_FLAG_THIN_LENS) != 0);
    int n = 0;
    for (int i = 0; i < pr->max_depth && n < max_seg; ++i) {
        g_ctx.bounce = (uint32_t)i;
        hitRecord rec; int prim;
        bool hit = world_hit(*h->w, r, pr->t_min, std::numeric_limits<float>::infinity(), rec, prim);
        float* q = rays + 6 * n; q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = r.dir.x; q[4] = r.dir.y; q[5] = r.dir.z;
        float* hh = hits + 3 * n; hh[0] = (float)prim; hh[1] = hit ? (float)rec.tri : -1.0f; hh[2] = hit ? rec.t : 0.0f;
        ++n;
        if (!hit) break;
        ray scattered; vec3 attenuation;
        if (!rec.matPtr->scatter(r, rec, attenuation, scattered)) break;
        r = scattered;
    }
    return n;
}

// Film::tonemap + Film::writeColour
void oracle_resolve_u8(const float* rgb, int64_t n_pixels, uint8_t* out) {
    for (int64_t i = 0; i < n_pixels; ++i) {
        vec3 c = tonemap(vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        writeColour(c, out + 3 * i);
    }
}
void oracle_tonemap(const float* rgb, int64_t n_pixels, float* out) {
    for (int64_t i = 0; i < n_pixels; ++i) {
        vec3 c = tonemap(vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
void oracle_sphere_uv(const float* p, float* uv) { getSphereUV(vec3(p[0], p[1], p[2]), uv[0], uv[1]); }

// shared math kernels on the CPU (same op codes as hrt_math_probe)
void oracle_math_probe(int32_t op, int64_t n, const float* in, const float* in2, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        switch (op) {
            case 0: out[i] = gsin(in[i]); break;
            case 1: out[i] = gcos(in[i]); break;
            case 2: out[i] = gacos(in[i]); break;
            case 3: out[i] = gatan2(in2[i], in[i]); break;
            case 4: out[i] = glog(in[i]); break;
            case 6: out[i] = in[i] * in2[i]; break;
            case 7: out[i] = in[i] / in2[i]; break;
            case 8: out[i] = sqrtf(in[i]); break;
            case 9: out[i] = in[i] + in2[i]; break;
            case 10: out[i] = fmaf(in[i], in2[i], in2[i]); break;
            case 11: out[i] = gsin_wide(in[i]); break;
            case 5: {
                u32x4 r = philox4x32_10(f2u(in[4 * i]), f2u(in[4 * i + 1]), f2u(in[4 * i + 2]), f2u(in[4 * i + 3]),
                                        f2u(in2[2 * i]), f2u(in2[2 * i + 1]));
                out[4 * i] = u2f(r.x); out[4 * i + 1] = u2f(r.y); out[4 * i + 2] = u2f(r.z); out[4 * i + 3] = u2f(r.w);
                break;
            }
            default: out[i] = 0.0f;
        }
    }
}
// distribution probes for the estimator-sanity tests
void oracle_spherical_rand(uint32_t seed, int64_t n, float* out) {
    rng_ctx c; c.seed_lo = seed; c.seed_hi = 0; c.sample = 0; c.bounce = 0;
    for (int64_t i = 0; i < n; ++i) {
        c.pixel = (uint32_t)i;
        u32x4 u = rng_draw(c, RNG_SCATTER, 0);
        vec3 v = spherical_rand(u.x, u.y);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}
void oracle_ball_rand(uint32_t seed, int64_t n, float* out) {
    rng_ctx c; c.seed_lo = seed; c.seed_hi = 0; c.sample = 0; c.bounce = 0;
    for (int64_t i = 0; i < n; ++i) {
        c.pixel = (uint32_t)i;
        vec3 v = ball_rand(c);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}
void oracle_quat_rotate_euler_deg(const float* euler_deg, const float* v, float* out) {
    quat q = quat_from_euler(vec3(gradians(euler_deg[0]), gradians(euler_deg[1]), gradians(euler_deg[2])));
    vec3 r = rotate(q, vec3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// glm::reflect / glm::refract / glm::normalize as restated in csrc/hrt_glm.h (material.h:168,224-225), for the KATs against
// independent float64 formulas (tests/test_oracle_kats.py): n vectors each, out 3 floats per vector
void oracle_reflect(int64_t n, const float* I, const float* N, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        vec3 r = reflect(vec3(I[3 * i], I[3 * i + 1], I[3 * i + 2]), vec3(N[3 * i], N[3 * i + 1], N[3 * i + 2]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void oracle_refract(int64_t n, const float* I, const float* N, const float* eta, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        vec3 r = refract(vec3(I[3 * i], I[3 * i + 1], I[3 * i + 2]), vec3(N[3 * i], N[3 * i + 1], N[3 * i + 2]), eta[i]);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void oracle_normalize(int64_t n, const float* v, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        vec3 r = normalize(vec3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}

}  // extern "C"
