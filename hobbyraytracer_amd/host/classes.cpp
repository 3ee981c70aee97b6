// classes.cpp — flatten() of every class of the reference's surface, Camera,
// Film and the mesh import seam.  See classes.h.
#include "classes.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "image_io.h"

namespace hrthost {

// ------------------------------------------------------------------ FlatBuilder
int FlatBuilder::addTexture(const std::shared_ptr<Texture>& t) {
    if (!t) throw FlattenError(HRT_ERR_INVALID, "null texture");
    auto it = tex_ids.find(t.get());
    if (it != tex_ids.end()) return it->second;
    const int id = (int)textures.size();
    tex_ids[t.get()] = id;
    textures.push_back(hrt_texture{});
    hrt_texture out{};
    t->flatten(*this, out);  // may append children (checker)
    textures[id] = out;
    return id;
}
int FlatBuilder::addMaterial(const std::shared_ptr<Material>& m) {
    if (!m) throw FlattenError(HRT_ERR_INVALID, "null material");
    auto it = mat_ids.find(m.get());
    if (it != mat_ids.end()) return it->second;
    const int id = (int)materials.size();
    mat_ids[m.get()] = id;
    materials.push_back(hrt_material{});
    hrt_material out{};
    out.albedo.tex = -1; out.s0.tex = -1; out.s1.tex = -1; out.mix_tex = -1;
    m->flatten(*this, out);
    materials[id] = out;
    return id;
}
int FlatBuilder::addMesh(const std::vector<float>& pos, const std::vector<float>& nrm, const std::vector<float>& uv,
                         const std::vector<float>& box, const std::vector<uint32_t>& refOrder, const std::vector<hrt_bvh_node>& nd) {
    hrt_mesh m{};
    m.tri_first = (uint32_t)(tri_pos.size() / 9);
    m.tri_count = (uint32_t)(pos.size() / 9);
    m.node_first = (uint32_t)nodes.size();
    m.node_count = (uint32_t)nd.size();
    tri_pos.insert(tri_pos.end(), pos.begin(), pos.end());
    tri_nrm.insert(tri_nrm.end(), nrm.begin(), nrm.end());
    tri_uv.insert(tri_uv.end(), uv.begin(), uv.end());
    tri_box.insert(tri_box.end(), box.begin(), box.end());
    tri_ref_order.insert(tri_ref_order.end(), refOrder.begin(), refOrder.end());
    nodes.insert(nodes.end(), nd.begin(), nd.end());
    meshes.push_back(m);
    return (int)meshes.size() - 1;
}
hrt_flat_scene FlatBuilder::flat() const {
    hrt_flat_scene f{};
    f.n_prims = (uint32_t)prims.size(); f.prims = prims.data();
    f.n_materials = (uint32_t)materials.size(); f.materials = materials.data();
    f.n_textures = (uint32_t)textures.size(); f.textures = textures.data();
    f.n_meshes = (uint32_t)meshes.size(); f.meshes = meshes.data();
    f.n_tris = tri_pos.size() / 9;
    f.tri_pos = tri_pos.data(); f.tri_nrm = tri_nrm.data(); f.tri_uv = tri_uv.data();
    f.tri_box = tri_box.size() == 6 * f.n_tris && f.n_tris ? tri_box.data() : nullptr;
    f.tri_ref_order = tri_ref_order.size() == f.n_tris && f.n_tris ? tri_ref_order.data() : nullptr;
    f.n_nodes = nodes.size(); f.nodes = nodes.data();
    f.n_texels_u8 = texels_u8.size(); f.texels_u8 = texels_u8.data();
    f.n_texels_f32 = texels_f32.size(); f.texels_f32 = texels_f32.data();
    f.background_tex = background;
    return f;
}

// ------------------------------------------------------------------ textures
void SolidColourTexture::flatten(FlatBuilder&, hrt_texture& out) const {
    out.kind = HRT_TEX_SOLID; out.c[0] = c.x; out.c[1] = c.y; out.c[2] = c.z;
}
void CheckeredTexture::flatten(FlatBuilder& fb, hrt_texture& out) const {
    out.kind = HRT_TEX_CHECKER;
    out.even = fb.addTexture(e);
    out.odd = fb.addTexture(o);
}
ImageTexture::ImageTexture(std::string filename) : width(0), height(0) {  // texture.cpp:30-51
    std::string err;
    if (!loadImageRGB8(filename, data, width, height, err)) {
        std::cout << "ERROR: Could not load image file: " << filename << std::endl;
        width = height = 0; data.clear();
    }
    std::cout << "Loaded image file: " << filename << std::endl;
}
void ImageTexture::flatten(FlatBuilder& fb, hrt_texture& out) const {
    out.kind = HRT_TEX_IMAGE; out.width = width; out.height = height; out.channels = 3;
    out.offset = data.empty() ? 0 : fb.addTexelsU8(data);
    if (data.empty()) out.width = out.height = 0;
}
EnvironmentMap::EnvironmentMap(std::string path) : width(0), height(0), channels(0) {  // texture.cpp:99-115
    std::string err;
    if (!loadImageF32(path, data, width, height, channels, err)) {
        std::cout << "ERROR: Could not environment map file: " << path << std::endl;
        width = height = channels = 0; data.clear();
        return;
    }
    std::cout << "Loaded environment map: " << path << std::endl;
}
void EnvironmentMap::flatten(FlatBuilder& fb, hrt_texture& out) const {
    out.kind = HRT_TEX_ENV; out.width = width; out.height = height; out.channels = channels;
    out.offset = data.empty() ? 0 : fb.addTexelsF32(data);
    if (data.empty()) out.width = out.height = 0;
}

// ------------------------------------------------------------------ materials
hrt_matvec3 MatVec3::flatten(FlatBuilder& fb) const {
    hrt_matvec3 m{}; m.tex = tex ? fb.addTexture(tex) : -1; m.c[0] = c.x; m.c[1] = c.y; m.c[2] = c.z; return m;
}
hrt_matscalar MatScalar::flatten(FlatBuilder& fb) const {
    hrt_matscalar m{}; m.tex = tex ? fb.addTexture(tex) : -1; m.c = c; return m;
}
void Isotropic::flatten(FlatBuilder& fb, hrt_material& out) const { out.kind = HRT_MAT_ISOTROPIC; out.albedo.tex = fb.addTexture(albedo); }
void DiffuseLight::flatten(FlatBuilder& fb, hrt_material& out) const { out.kind = HRT_MAT_DIFFUSE_LIGHT; out.albedo = emit.flatten(fb); out.s0 = s.flatten(fb); }
void UVTest::flatten(FlatBuilder&, hrt_material& out) const { out.kind = HRT_MAT_UVTEST; }
void Lambertian::flatten(FlatBuilder& fb, hrt_material& out) const { out.kind = HRT_MAT_LAMBERTIAN; out.albedo = albedo.flatten(fb); }
void Metal::flatten(FlatBuilder& fb, hrt_material& out) const { out.kind = HRT_MAT_METAL; out.albedo = albedo.flatten(fb); out.s0 = r.flatten(fb); }
void Dielectric::flatten(FlatBuilder& fb, hrt_material& out) const { out.kind = HRT_MAT_DIELECTRIC; out.s0 = ir.flatten(fb); out.s1 = r.flatten(fb); }
void PBR::flatten(FlatBuilder& fb, hrt_material& out) const {
    out.kind = HRT_MAT_PBR; out.albedo = alb.flatten(fb); out.s0.tex = -1; out.s0.c = rough; out.mix_tex = fb.addTexture(mix);
}

// ------------------------------------------------------------------ hittables
namespace {
hrt_prim basePrim(FlatBuilder& fb, int kind, const std::shared_ptr<Material>& m) {
    hrt_prim p{};
    p.kind = kind; p.material = fb.addMaterial(m); p.mesh = -1; p.boundary_kind = -1; p.density = 0.0f;
    if (fb.chain.size() > HRT_MAX_XFORMS) throw FlattenError(HRT_ERR_UNSUPPORTED, "more than HRT_MAX_XFORMS nested instance wrappers");
    p.n_xforms = (int32_t)fb.chain.size();
    for (size_t k = 0; k < fb.chain.size(); ++k) p.xf[k] = fb.chain[k];
    return p;
}
struct ChainPush {
    FlatBuilder& fb;
    ChainPush(FlatBuilder& f, const hrt_xform& x) : fb(f) { fb.chain.push_back(x); }
    ~ChainPush() { fb.chain.pop_back(); }
};
}  // namespace

void HittableList::flatten(FlatBuilder& fb) const {
    // scene.cpp:376-379: the world is a HittableList; nested lists flatten in place
    // (HittableList::hit is associative over closest-so-far when there is no wrapper in between)
    if (!fb.chain.empty()) throw FlattenError(HRT_ERR_UNSUPPORTED, "a HittableList inside an instance wrapper is not supported");
    for (const auto& o : objects) {
        if (!o) throw FlattenError(HRT_ERR_INVALID, "null object in the world list (unknown object type? scene.cpp:279,356)");
        o->flatten(fb);
    }
}
void Sphere::flatten(FlatBuilder& fb) const { hrt_prim p = basePrim(fb, HRT_PRIM_SPHERE, matPtr); params(p.p); fb.addPrim(p); }
void Triangle::flatten(FlatBuilder& fb) const {
    hrt_prim p = basePrim(fb, HRT_PRIM_TRIANGLE, matPtr);
    p.p[0] = v0.x; p.p[1] = v0.y; p.p[2] = v0.z; p.p[3] = v1.x; p.p[4] = v1.y; p.p[5] = v1.z; p.p[6] = v2.x; p.p[7] = v2.y; p.p[8] = v2.z;
    fb.addPrim(p);
}
void YZRect::flatten(FlatBuilder& fb) const { hrt_prim p = basePrim(fb, HRT_PRIM_YZ_RECT, mp); p.p[0] = y0; p.p[1] = y1; p.p[2] = z0; p.p[3] = z1; p.p[4] = k; fb.addPrim(p); }
void XZRect::flatten(FlatBuilder& fb) const { hrt_prim p = basePrim(fb, HRT_PRIM_XZ_RECT, mp); p.p[0] = x0; p.p[1] = x1; p.p[2] = z0; p.p[3] = z1; p.p[4] = k; fb.addPrim(p); }
void XYRect::flatten(FlatBuilder& fb) const { hrt_prim p = basePrim(fb, HRT_PRIM_XY_RECT, mp); p.p[0] = x0; p.p[1] = x1; p.p[2] = y0; p.p[3] = y1; p.p[4] = k; fb.addPrim(p); }
void Box::flatten(FlatBuilder& fb) const { hrt_prim p = basePrim(fb, HRT_PRIM_BOX, matPtr); params(p.p); fb.addPrim(p); }

void Translate::flatten(FlatBuilder& fb) const {
    hrt_xform x{}; x.kind = HRT_XF_TRANSLATE; x.v[0] = offset.x; x.v[1] = offset.y; x.v[2] = offset.z;
    ChainPush g(fb, x); ptr->flatten(fb);
}
void Scale::flatten(FlatBuilder& fb) const {
    hrt_xform x{}; x.kind = HRT_XF_SCALE; x.v[0] = factor.x; x.v[1] = factor.y; x.v[2] = factor.z;
    ChainPush g(fb, x); ptr->flatten(fb);
}
void RotateQuat::flatten(FlatBuilder& fb) const {
    hrt_xform x{}; x.kind = HRT_XF_ROTATE_QUAT; x.v[0] = rotation.x; x.v[1] = rotation.y; x.v[2] = rotation.z; x.v[3] = rotation.w;
    ChainPush g(fb, x); ptr->flatten(fb);
}
RotateY::RotateY(std::shared_ptr<Hittable> p, float angle) : ptr(p) {  // rotateY.cpp:4-9
    float radians = hrt::gradians(angle);
    sinTheta = hrt::gsin(radians);
    cosTheta = hrt::gcos(radians);
}
void RotateY::flatten(FlatBuilder& fb) const {
    hrt_xform x{}; x.kind = HRT_XF_ROTATE_Y; x.v[0] = sinTheta; x.v[1] = cosTheta;
    ChainPush g(fb, x); ptr->flatten(fb);
}
void ConstantMedium::flatten(FlatBuilder& fb) const {
    hrt_prim p = basePrim(fb, HRT_PRIM_MEDIUM, phaseFunction);
    if (auto s = std::dynamic_pointer_cast<Sphere>(boundary)) { p.boundary_kind = HRT_PRIM_SPHERE; s->params(p.p); }
    else if (auto b = std::dynamic_pointer_cast<Box>(boundary)) { p.boundary_kind = HRT_PRIM_BOX; b->params(p.p); }
    else throw FlattenError(HRT_ERR_UNSUPPORTED, "ConstantMedium boundary must be a Sphere or a Box (a mesh boundary never scatters in the reference: SURVEY Q-5)");
    p.density = density;
    fb.addPrim(p);
}

// ------------------------------------------------------------------ Mesh (mesh.cpp:13-51)
Mesh::Mesh(std::string filepath, std::shared_ptr<Material> m) : matPtr(m) {
    std::string err;
    ok = importFile(filepath, soup, err);
    if (!ok) {
        std::cerr << "Assimp error: " << err << std::endl;  // message format of mesh.cpp:59
        soup = TriangleSoup();
    } else {
        std::cout << "Loaded mesh: " << filepath << std::endl;
    }
    tree = std::make_shared<BVHNode>(soup);
    std::cout << "Indexed file: " << filepath << std::endl;
}
Mesh::Mesh(TriangleSoup s, std::shared_ptr<Material> m) : soup(std::move(s)), matPtr(m), ok(true) {
    tree = std::make_shared<BVHNode>(soup);
}
void Mesh::flatten(FlatBuilder& fb) const {
    hrt_prim p = basePrim(fb, HRT_PRIM_MESH, matPtr);
    p.mesh = fb.addMesh(soup.pos, soup.nrm, soup.uv, tree->leafBoxes, tree->refOrder, tree->nodes);
    fb.addPrim(p);
}

// The import seam.  The reference calls Assimp::Importer::ReadFile(path,
// aiProcess_Triangulate | aiProcess_FlipUVs) (mesh.cpp:56) and un-indexes the
// result.  Assimp is an empty submodule in the snapshot, so Wavefront OBJ —
// the only format the sample scenes use — is parsed here with the same
// observable result: polygons are fan-triangulated, v -> 1 - v (FlipUVs), no
// normals are generated (missing normals become (0,0,0): mesh.cpp:83-90, Q-9),
// missing UVs become (0,0) (mesh.cpp:92-99).
// Q-8 (mesh.cpp:111-114): the reference appends every aiMesh's face indices to one global list WITHOUT adding the
// number of vertices already there.  Assimp's OBJ importer makes one aiMesh per object / group / material run that has
// faces and, without aiProcess_JoinIdenticalVertices, one vertex per face corner, so sub-mesh k's indices are
// 0 .. 3 F_k - 1: its F_k triangles come out as copies of the FIRST F_k triangles of the whole file.  That is the
// default here too (Mesh::objIndexQuirk, SURVEY 8.1: "default = reference"); single-object files -- every asset this
// repository generates -- are not affected.  Switch: HRT_OBJ_INDICES=rebased in the environment, or the CLI's
// --obj-indices rebased, read the file as one mesh with correct indices.
bool Mesh::objIndexQuirk() {
    const char* e = std::getenv("HRT_OBJ_INDICES");
    return !(e && std::string(e) == "rebased");
}
bool Mesh::importFile(const std::string& path, TriangleSoup& out, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "Unable to open file \"" + path + "\"."; return false; }
    std::string lower = path;
    for (char& c : lower) c = (char)std::tolower((unsigned char)c);
    if (lower.size() < 4 || lower.compare(lower.size() - 4, 4, ".obj") != 0) { err = "No suitable reader found for the file format of file \"" + path + "\"."; return false; }
    std::vector<float> v, vt, vn;
    std::string line;
    struct Idx { int v, t, n; };
    std::vector<Idx> face;
    std::vector<size_t> subMeshCorner;  // first corner of every aiMesh Assimp would make (object / group / material runs with faces)
    std::vector<float> cPos, cNrm, cUv; // one vertex per face corner, in file order (= the aiMeshes' vertex arrays, concatenated: mesh.cpp:108-110)
    std::vector<uint32_t> triIdx, triSub;   // the triangulated faces: three corner indices each, and the sub-mesh they belong to
    bool newRun = true;
    int lineNo = 0;
    while (std::getline(f, line)) {
        ++lineNo;
        // line continuation
        while (!line.empty() && (line.back() == '\r')) line.pop_back();
        while (!line.empty() && line.back() == '\\') {
            line.pop_back();
            std::string more;
            if (!std::getline(f, more)) break;
            line += " " + more;
        }
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float x = 0, y = 0, z = 0;
            if (std::sscanf(s + 2, "%f %f %f", &x, &y, &z) < 3) { err = "OBJ: bad vertex at line " + std::to_string(lineNo); return false; }
            if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) { err = "OBJ: non-finite vertex at line " + std::to_string(lineNo); return false; }
            v.push_back(x); v.push_back(y); v.push_back(z);
        } else if (s[0] == 'v' && s[1] == 't') {
            float a = 0, b = 0;
            int k = std::sscanf(s + 2, "%f %f", &a, &b);
            if (k < 1) { err = "OBJ: bad texcoord at line " + std::to_string(lineNo); return false; }
            vt.push_back(a); vt.push_back(b);
        } else if (s[0] == 'v' && s[1] == 'n') {
            float x = 0, y = 0, z = 0;
            if (std::sscanf(s + 2, "%f %f %f", &x, &y, &z) < 3) { err = "OBJ: bad normal at line " + std::to_string(lineNo); return false; }
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            const char* p = s + 1;
            for (;;) {
                while (*p == ' ' || *p == '\t') ++p;
                if (!*p) break;
                Idx ix{0, 0, 0};
                char* e;
                ix.v = (int)std::strtol(p, &e, 10);
                if (e == p) { err = "OBJ: bad face at line " + std::to_string(lineNo); return false; }
                p = e;
                if (*p == '/') {
                    ++p;
                    if (*p != '/') { ix.t = (int)std::strtol(p, &e, 10); p = e; }
                    if (*p == '/') { ++p; ix.n = (int)std::strtol(p, &e, 10); p = e; }
                }
                auto fix = [](int i, size_t count) { return i < 0 ? (int)count + i : i - 1; };
                ix.v = fix(ix.v, v.size() / 3);
                ix.t = ix.t ? fix(ix.t, vt.size() / 2) : -1;
                ix.n = ix.n ? fix(ix.n, vn.size() / 3) : -1;
                if (ix.v < 0 || (size_t)ix.v >= v.size() / 3 || (ix.t >= 0 && (size_t)ix.t >= vt.size() / 2) ||
                    (ix.n >= 0 && (size_t)ix.n >= vn.size() / 3) || ix.t < -1 || ix.n < -1) {
                    err = "OBJ: index out of range at line " + std::to_string(lineNo); return false;
                }
                face.push_back(ix);
            }
            if (face.size() < 3) continue;  // points / lines are dropped by Triangulate
            if (newRun) { subMeshCorner.push_back(cPos.size() / 3); newRun = false; }
            // Assimp's OBJ importer makes one vertex per face CORNER (no aiProcess_JoinIdenticalVertices, mesh.cpp:56) and
            // aiProcess_Triangulate then indexes those: a polygon's corners c0..ck-1 become the fan (c0, ci, ci+1) -- its choice for a
            // convex polygon of up to four corners; larger or concave ones are ear-clipped there, which covers the same area
            const uint32_t c0 = (uint32_t)(cPos.size() / 3);
            for (const Idx& ix : face) {
                cPos.push_back(v[3 * ix.v]); cPos.push_back(v[3 * ix.v + 1]); cPos.push_back(v[3 * ix.v + 2]);
                if (ix.n >= 0) { cNrm.push_back(vn[3 * ix.n]); cNrm.push_back(vn[3 * ix.n + 1]); cNrm.push_back(vn[3 * ix.n + 2]); }
                else { cNrm.push_back(0); cNrm.push_back(0); cNrm.push_back(0); }
                if (ix.t >= 0) { cUv.push_back(vt[2 * ix.t]); cUv.push_back(1.0f - vt[2 * ix.t + 1]); }      // aiProcess_FlipUVs
                else { cUv.push_back(0); cUv.push_back(0); }
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                triIdx.push_back(c0); triIdx.push_back(c0 + (uint32_t)k); triIdx.push_back(c0 + (uint32_t)k + 1);
                triSub.push_back((uint32_t)subMeshCorner.size() - 1);
            }
        }
        else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || s[1] == 0)) newRun = true;
        else if (std::strncmp(s, "usemtl", 6) == 0 && (s[6] == ' ' || s[6] == '\t')) newRun = true;
    }
    if (triIdx.empty()) { err = "OBJ: file contains no faces: " + path; return false; }
    // Q-8 (mesh.cpp:111-114): every aiMesh's indices are local to ITS vertex array, and the reference appends them to one list
    // without adding the vertices already there -- so the corners of sub-mesh m > 0 are looked up at the START of the concatenated
    // vertex array (the corners of the first sub-meshes), vertex by vertex: a file of triangles only shows whole triangles of the
    // first sub-mesh again, one with quads mixes corners of different faces.  Reproduced at that level: local index = corner index
    // minus the sub-mesh's first corner.
    const bool quirk = objIndexQuirk() && subMeshCorner.size() > 1;
    const size_t nTri = triSub.size();
    out.pos.resize(9 * nTri); out.nrm.resize(9 * nTri); out.uv.resize(6 * nTri);
    for (size_t t = 0; t < nTri; ++t)
        for (int k = 0; k < 3; ++k) {
            size_t c = triIdx[3 * t + k];
            if (quirk) c -= subMeshCorner[triSub[t]];      // < the sub-mesh's own corner count <= all corners: always in range
            for (int d = 0; d < 3; ++d) { out.pos[9 * t + 3 * k + d] = cPos[3 * c + d]; out.nrm[9 * t + 3 * k + d] = cNrm[3 * c + d]; }
            out.uv[6 * t + 2 * k] = cUv[2 * c]; out.uv[6 * t + 2 * k + 1] = cUv[2 * c + 1];
        }
    return true;
}

// ------------------------------------------------------------------ Camera (camera.h:9-27)
Camera::Camera(vec3 lookFrom, vec3 lookAt, vec3 up, float vfov, float aspectRatio, float aperture, float focusDistance) {
    float theta = hrt::gradians(vfov);
    float h = std::tan(theta / 2);  // setup-time only: the resulting constants are shared by oracle and device
    float viewportHeight = 2.0f * h;
    float viewportWidth = aspectRatio * viewportHeight;
    w = hrt::normalize(lookFrom - lookAt);
    u = hrt::normalize(hrt::cross(up, w));
    v = hrt::cross(w, u);
    origin = lookFrom;
    horizontal = focusDistance * viewportWidth * u;
    vertical = focusDistance * viewportHeight * v;
    lowerLeftCorner = origin - horizontal / 2.0f - vertical / 2.0f - focusDistance * w;
    lensRadius = aperture / 2.0f;
}
hrt_camera Camera::flatten() const {
    hrt_camera c{};
    c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
    c.lower_left[0] = lowerLeftCorner.x; c.lower_left[1] = lowerLeftCorner.y; c.lower_left[2] = lowerLeftCorner.z;
    c.horizontal[0] = horizontal.x; c.horizontal[1] = horizontal.y; c.horizontal[2] = horizontal.z;
    c.vertical[0] = vertical.x; c.vertical[1] = vertical.y; c.vertical[2] = vertical.z;
    c.lens_u[0] = u.x; c.lens_u[1] = u.y; c.lens_u[2] = u.z;
    c.lens_v[0] = v.x; c.lens_v[1] = v.y; c.lens_v[2] = v.z;
    c.lens_radius = lensRadius;
    return c;
}

// ------------------------------------------------------------------ Film (film.cpp:18-23, 59-79)
Film::Film(int w, int h, int samples, std::string output) : outputName(output) { resize(w, h, samples); }
void Film::resize(int w, int h, int samples) {
    f = {w, h, samples};
    pixels.assign((size_t)w * h * 3, 0);
    lin.assign((size_t)w * h * 3, 0.0f);
}
int Film::outputFilm() {
    auto ends = [&](const char* suf) { std::string s(suf); return outputName.size() >= s.size() && outputName.compare(outputName.size() - s.size(), s.size(), s) == 0; };
    if (ends(".png")) return writePNG(outputName, pixels.data(), f.width, f.height, f.width * 3) ? 1 : 0;
    if (ends(".tga")) return writeTGA(outputName, pixels.data(), f.width, f.height) ? 1 : 0;
    if (!ends(".bmp")) std::cout << "File type not supported, generating bitmap!" << std::endl;
    std::cout << ">>> " << outputName << std::endl;
    return writeBMP(outputName, pixels.data(), f.width, f.height) ? 1 : 0;
}

void flattenWorld(FlatBuilder& fb, const std::shared_ptr<Hittable>& world, const std::shared_ptr<Texture>& background) {
    fb.chain.clear();
    fb.setBackground(background);
    world->flatten(fb);
}

}  // namespace hrthost
