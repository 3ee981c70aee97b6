// classes.h — the reference's Hittable / Material / Texture / Camera / Film
// class surface, kept on the host so that scene construction code written
// against the reference (scene.cpp:239-357, or programmatic scenes) still
// reads the same.  The classes no longer intersect anything themselves: each
// gains `flatten()` (SURVEY.md §7.1 step 2 — the reference's members are
// private with no getters, so introspection has to live inside the classes),
// which emits the SoA `hrt_flat_scene` consumed by libhrt_hip.so.
//
// Reference anchors: hittable.h:27-32, hittableList.h, sphere.h, aarect.h,
// box.h, triangle.h, mesh.h, bvh.h, translate.h, scale.h, rotateQuat.h,
// rotateY.h, constantMedium.h, material.h, texture.h, camera.h, film.h.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hrt.h"
#include "../csrc/hrt_glm.h"

namespace hrthost {

using hrt::quat;
using hrt::vec2;
using hrt::vec3;

class Texture;
class Material;
class Hittable;

struct FlattenError : std::runtime_error {
    hrt_status status;
    FlattenError(hrt_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// Collects the flattened arrays; de-duplicates shared textures / materials by
// object identity (the reference shares them through shared_ptr).
class FlatBuilder {
public:
    int addTexture(const std::shared_ptr<Texture>& t);
    int addMaterial(const std::shared_ptr<Material>& m);
    // appends a triangle soup + its flattened BVH; returns the mesh index
    int addMesh(const std::vector<float>& pos, const std::vector<float>& nrm, const std::vector<float>& uv,
                const std::vector<float>& box, const std::vector<uint32_t>& refOrder, const std::vector<hrt_bvh_node>& nodes);
    void addPrim(const hrt_prim& p) { prims.push_back(p); }
    uint64_t addTexelsU8(const std::vector<uint8_t>& d) { uint64_t o = texels_u8.size(); texels_u8.insert(texels_u8.end(), d.begin(), d.end()); return o; }
    uint64_t addTexelsF32(const std::vector<float>& d) { uint64_t o = texels_f32.size(); texels_f32.insert(texels_f32.end(), d.begin(), d.end()); return o; }
    void setBackground(const std::shared_ptr<Texture>& t) { background = addTexture(t); }

    // the finished scene; pointers stay valid while the builder lives and is not modified
    hrt_flat_scene flat() const;

    std::vector<hrt_prim> prims;
    std::vector<hrt_material> materials;
    std::vector<hrt_texture> textures;
    std::vector<hrt_mesh> meshes;
    std::vector<float> tri_pos, tri_nrm, tri_uv, tri_box;
    std::vector<uint32_t> tri_ref_order;
    std::vector<hrt_bvh_node> nodes;
    std::vector<uint8_t> texels_u8;
    std::vector<float> texels_f32;
    int background = -1;

    std::vector<hrt_xform> chain;  // wrapper chain being flattened, outermost first
private:
    std::map<const Texture*, int> tex_ids;
    std::map<const Material*, int> mat_ids;
};

// ---------------------------------------------------------------- texture.h
class Texture {
public:
    virtual ~Texture() {}
    virtual void flatten(FlatBuilder& fb, hrt_texture& out) const = 0;
};
class SolidColourTexture : public Texture {
public:
    SolidColourTexture() : c(0.0f) {}
    SolidColourTexture(vec3 colour) : c(colour) {}
    SolidColourTexture(float f) : c(vec3(f)) {}
    SolidColourTexture(float r, float g, float b) : c(vec3(r, g, b)) {}
    void flatten(FlatBuilder&, hrt_texture& out) const override;
private:
    vec3 c;
};
class CheckeredTexture : public Texture {
public:
    CheckeredTexture(std::shared_ptr<Texture> even, std::shared_ptr<Texture> odd) : e(even), o(odd) {}
    CheckeredTexture(vec3 colour1, vec3 colour2)
        : e(std::make_shared<SolidColourTexture>(colour1)), o(std::make_shared<SolidColourTexture>(colour2)) {}
    void flatten(FlatBuilder& fb, hrt_texture& out) const override;
private:
    std::shared_ptr<Texture> e, o;
};
class ImageTexture : public Texture {  // texture.cpp:30-51: stbi_load forced to 3 channels
public:
    const static int bytesPerPixel = 3;
    ImageTexture() : width(0), height(0) {}
    ImageTexture(std::string filename);
    ImageTexture(std::vector<uint8_t> rgb, int w, int h) : data(std::move(rgb)), width(w), height(h) {}
    void flatten(FlatBuilder& fb, hrt_texture& out) const override;
private:
    std::vector<uint8_t> data;
    int width, height;
};
class EnvironmentMap : public Texture {  // texture.cpp:99-115: stbi_loadf, native channel count
public:
    EnvironmentMap() : width(0), height(0), channels(0) {}
    EnvironmentMap(std::string path);
    EnvironmentMap(std::vector<float> d, int w, int h, int ch) : data(std::move(d)), width(w), height(h), channels(ch) {}
    void flatten(FlatBuilder& fb, hrt_texture& out) const override;
private:
    std::vector<float> data;
    int width, height, channels;
};

// ---------------------------------------------------------------- material.h
class MatVec3 {  // material.h:10-35
public:
    MatVec3(vec3 v) : c(v) {}
    MatVec3(std::shared_ptr<Texture> t) : c(0.0f), tex(t) {}
    hrt_matvec3 flatten(FlatBuilder& fb) const;
private:
    vec3 c;
    std::shared_ptr<Texture> tex;
};
class MatScalar {  // material.h:37-58
public:
    MatScalar(float v) : c(v) {}
    MatScalar(std::shared_ptr<Texture> t) : c(0.0f), tex(t) {}
    hrt_matscalar flatten(FlatBuilder& fb) const;
private:
    float c;
    std::shared_ptr<Texture> tex;
};
class Material {
public:
    virtual ~Material() {}
    virtual void flatten(FlatBuilder& fb, hrt_material& out) const = 0;
};
class Isotropic : public Material {
public:
    Isotropic(vec3 c) : albedo(std::make_shared<SolidColourTexture>(c)) {}
    Isotropic(std::shared_ptr<Texture> a) : albedo(a) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    std::shared_ptr<Texture> albedo;
};
class DiffuseLight : public Material {
public:
    DiffuseLight(MatVec3 colour, MatScalar strength) : emit(colour), s(strength) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    MatVec3 emit; MatScalar s;
};
class UVTest : public Material {
public:
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
};
class Lambertian : public Material {
public:
    Lambertian(MatVec3 a) : albedo(a) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    MatVec3 albedo;
};
class Metal : public Material {
public:
    Metal(MatVec3 colour, MatScalar roughness) : albedo(colour), r(roughness) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    MatVec3 albedo; MatScalar r;
};
class Dielectric : public Material {
public:
    Dielectric(MatScalar indexOfRefraction, MatScalar roughness) : ir(indexOfRefraction), r(roughness) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    MatScalar ir, r;
};
class PBR : public Material {  // material.cpp:4-16
public:
    PBR(vec3 albedo, float metallness, float roughness)
        : alb(albedo), rough(roughness), mix(std::make_shared<SolidColourTexture>(metallness)) {}
    PBR(std::shared_ptr<Texture> albedo, std::shared_ptr<Texture> metallness, float roughness)
        : alb(albedo), rough(roughness), mix(metallness) {}
    void flatten(FlatBuilder& fb, hrt_material& out) const override;
private:
    MatVec3 alb; float rough; std::shared_ptr<Texture> mix;
};

// ---------------------------------------------------------------- hittable.h
class Hittable {
public:
    virtual ~Hittable() {}
    // Emits this object (under fb.chain) into the flat scene.
    virtual void flatten(FlatBuilder& fb) const = 0;
};

class HittableList : public Hittable {  // hittableList.h
public:
    HittableList() {}
    HittableList(std::shared_ptr<Hittable> object) { add(object); }
    void clear() { objects.clear(); }
    void add(std::shared_ptr<Hittable> object) { objects.push_back(object); }
    void flatten(FlatBuilder& fb) const override;
    std::vector<std::shared_ptr<Hittable>> objects;
};

class Sphere : public Hittable {
public:
    Sphere(vec3 c, float r, std::shared_ptr<Material> m) : center(c), radius(r), matPtr(m) {}
    void flatten(FlatBuilder& fb) const override;
    void params(float* p) const { p[0] = center.x; p[1] = center.y; p[2] = center.z; p[3] = radius; }
private:
    vec3 center; float radius; std::shared_ptr<Material> matPtr;
};
class Triangle : public Hittable {  // triangle.h:6-19 (the stand-alone triangle; meshes are ITriangle soups)
public:
    Triangle() : v0(0.0f), v1(0.0f), v2(0.0f), matPtr(nullptr) {}
    Triangle(vec3 _v0, vec3 _v1, vec3 _v2, std::shared_ptr<Material> m) : v0(_v0), v1(_v1), v2(_v2), matPtr(m) {}
    void flatten(FlatBuilder& fb) const override;
private:
    vec3 v0, v1, v2; std::shared_ptr<Material> matPtr;
};
class YZRect : public Hittable {
public:
    YZRect(float _y0, float _y1, float _z0, float _z1, float _k, std::shared_ptr<Material> m) : y0(_y0), y1(_y1), z0(_z0), z1(_z1), k(_k), mp(m) {}
    void flatten(FlatBuilder& fb) const override;
private:
    float y0, y1, z0, z1, k; std::shared_ptr<Material> mp;
};
class XZRect : public Hittable {
public:
    XZRect(float _x0, float _x1, float _z0, float _z1, float _k, std::shared_ptr<Material> m) : x0(_x0), x1(_x1), z0(_z0), z1(_z1), k(_k), mp(m) {}
    void flatten(FlatBuilder& fb) const override;
private:
    float x0, x1, z0, z1, k; std::shared_ptr<Material> mp;
};
class XYRect : public Hittable {
public:
    XYRect(float _x0, float _x1, float _y0, float _y1, float _k, std::shared_ptr<Material> m) : x0(_x0), x1(_x1), y0(_y0), y1(_y1), k(_k), mp(m) {}
    void flatten(FlatBuilder& fb) const override;
private:
    float x0, x1, y0, y1, k; std::shared_ptr<Material> mp;
};
class Box : public Hittable {  // box.h
public:
    Box(vec3 center, vec3 dimensions, std::shared_ptr<Material> m)
        : boxMin(center - dimensions / 2.0f), boxMax(center + dimensions / 2.0f), matPtr(m) {}
    static std::shared_ptr<Box> minMaxBox(vec3 mn, vec3 mx, std::shared_ptr<Material> m) {
        return std::make_shared<Box>((mn + mx) / vec3(2.0f), mx - mn, m);
    }
    void flatten(FlatBuilder& fb) const override;
    void params(float* p) const { p[0] = boxMin.x; p[1] = boxMin.y; p[2] = boxMin.z; p[3] = boxMax.x; p[4] = boxMax.y; p[5] = boxMax.z; }
private:
    vec3 boxMin, boxMax; std::shared_ptr<Material> matPtr;
};

// ITriangle soup of one mesh (triangle.h:20-37): 9 / 9 / 6 floats per triangle.
struct TriangleSoup {
    std::vector<float> pos, nrm, uv;
    size_t size() const { return pos.size() / 9; }
};

// bvh.h — the flattened replacement of the pointer tree: 64-byte two-child
// nodes over a triangle soup that is reordered into leaf order.
// hrt_host_set_bvh_builder (include/hrt_host.h): fn = hrt_host_bvh_build_fn or nullptr
void setDeviceBvhBuilder(void* fn, int device);
class BVHNode : public Hittable {
public:
    BVHNode() {}
    // Builds the flattened BVH; `soup` is reordered in place.
    explicit BVHNode(TriangleSoup& soup);
    void flatten(FlatBuilder&) const override { throw FlattenError(HRT_ERR_UNSUPPORTED, "BVHNode is flattened through its Mesh"); }
    std::vector<hrt_bvh_node> nodes;
    std::vector<float> leafBoxes;  // 6 floats per triangle: reference leaf-level boxes (hrt_flat_scene::tri_box)
    std::vector<uint32_t> refOrder; // per triangle: position in the reference tree's leaf order (tri_ref_order)
    int depth = 0;
};

class Mesh : public Hittable {  // mesh.h
public:
    Mesh(std::string filepath, std::shared_ptr<Material> matPtr);
    Mesh(TriangleSoup soup, std::shared_ptr<Material> matPtr);
    void flatten(FlatBuilder& fb) const override;
    size_t triangleCount() const { return soup.size(); }
    bool loaded() const { return ok; }
private:
    // The importer seam (mesh.cpp:53-120 calls Assimp with Triangulate|FlipUVs).
    static bool importFile(const std::string& path, TriangleSoup& out, std::string& err);
    // Q-8 (mesh.cpp:111-114, un-rebased indices of multi-mesh files): true = the reference's behaviour (default);
    // HRT_OBJ_INDICES=rebased in the environment turns it off.
    static bool objIndexQuirk();
    TriangleSoup soup;
    std::shared_ptr<BVHNode> tree;
    std::shared_ptr<Material> matPtr;
    bool ok = false;
};

class Translate : public Hittable {
public:
    Translate(std::shared_ptr<Hittable> object, const vec3& displacement) : ptr(object), offset(displacement) {}
    void flatten(FlatBuilder& fb) const override;
private:
    std::shared_ptr<Hittable> ptr; vec3 offset;
};
class Scale : public Hittable {
public:
    Scale(std::shared_ptr<Hittable> object, vec3 f) : ptr(object), factor(f) {}
    void flatten(FlatBuilder& fb) const override;
private:
    std::shared_ptr<Hittable> ptr; vec3 factor;
};
class RotateQuat : public Hittable {
public:
    RotateQuat(std::shared_ptr<Hittable> object, quat r) : ptr(object), rotation(r) {}
    void flatten(FlatBuilder& fb) const override;
private:
    std::shared_ptr<Hittable> ptr; quat rotation;
};
class RotateY : public Hittable {
public:
    RotateY(std::shared_ptr<Hittable> p, float angle);  // rotateY.cpp:4-9
    void flatten(FlatBuilder& fb) const override;
private:
    std::shared_ptr<Hittable> ptr; float sinTheta, cosTheta;
};
class ConstantMedium : public Hittable {  // constantMedium.h
public:
    ConstantMedium(std::shared_ptr<Hittable> b, float d, std::shared_ptr<Texture> a)
        : boundary(b), density(d), phaseFunction(std::make_shared<Isotropic>(a)) {}
    ConstantMedium(std::shared_ptr<Hittable> b, float d, vec3 colour)
        : boundary(b), density(d), phaseFunction(std::make_shared<Isotropic>(colour)) {}
    void flatten(FlatBuilder& fb) const override;
private:
    std::shared_ptr<Hittable> boundary; float density; std::shared_ptr<Material> phaseFunction;
};

// ---------------------------------------------------------------- camera.h:9-39
class Camera {
public:
    Camera() {}
    Camera(vec3 lookFrom, vec3 lookAt, vec3 up, float vfov, float aspectRatio, float aperture, float focusDistance);
    hrt_camera flatten() const;
private:
    vec3 origin, lowerLeftCorner, horizontal, vertical, w, u, v;
    float lensRadius = 0.0f;
};

// ---------------------------------------------------------------- film.h
struct film_desc { int width, height, samples; };
class Film {
public:
    Film(int w, int h, int samples, std::string output);
    film_desc getFilm() const { return f; }
    float getAspectRatio() const { return (float)f.width / (float)f.height; }
    uint8_t* getPixels() { return pixels.data(); }
    std::vector<float>& linear() { return lin; }  // fp32 linear means (for exact diffs / RMSE)
    const std::string& output() const { return outputName; }
    void setOutput(const std::string& o) { outputName = o; }
    void resize(int w, int h, int samples);
    int outputFilm();  // film.cpp:59-79: 1 on success (stb convention, Q-12)
private:
    std::vector<uint8_t> pixels;
    std::vector<float> lin;
    film_desc f;
    std::string outputName;
};

// ---------------------------------------------------------------- scene.h
class Scene {
public:
    Scene() : isLoaded(false) {}
    // scene.cpp:127-374.  Returns 1 on success, -1 on failure (message on stdout).
    // `assetDir` is prepended to relative mesh / image paths that do not exist
    // relative to the cwd (the reference resolves against the cwd only).
    int loadScene(std::string path, std::string assetDir = "");
    std::shared_ptr<HittableList> getScene() { return std::make_shared<HittableList>(objects); }
    const Camera& getCamera() const { return camera; }
    const std::shared_ptr<Texture>& getBackground() const { return background; }
    const std::shared_ptr<Film>& getFilm() const { return film; }
    // rebuilds the camera for an overridden film aspect (CLI --size)
    void setFilmSize(int w, int h, int samples);
    std::string lastError;
private:
    std::map<std::string, std::shared_ptr<Material>> materials;
    std::map<std::string, std::shared_ptr<Texture>> textures;
    HittableList objects;
    Camera camera;
    std::shared_ptr<Texture> background;
    std::shared_ptr<Film> film;
    struct CamDesc { vec3 position, lookAt, up; float fov = 0, aperture = 0, focus = 0; } camDesc;
    std::string assetDir;
    std::string resolve(const std::string& p) const;
    bool isLoaded;
};

// flattens world + background in one go
void flattenWorld(FlatBuilder& fb, const std::shared_ptr<Hittable>& world, const std::shared_ptr<Texture>& background);

}  // namespace hrthost
