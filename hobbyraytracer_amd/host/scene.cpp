// scene.cpp — Scene::loadScene, following scene.cpp:127-374 of the reference
// key by key (schema: SURVEY.md §5.6), on top of yaml_lite instead of yaml-cpp.
//
// Same acceptance set as the reference plus ADDITIVE keys that make the classes
// the reference cannot reach from YAML loadable:
//   materials: dielectric{ior, roughness}, isotropic, pbr{metalness, roughness}, uv_test
//   objects:   box{min,max | center,dimensions}, constant_medium{boundary{sphere|box}, density, colour}, triangle{v0,v1,v2}
//   transform: rotate_y: degrees (innermost)
// Behavioural differences, all on error paths: an unknown object `type` is
// reported and fails the load (the reference pushes a nullptr into the world
// and crashes at render time, scene.cpp:279,356); a mesh that fails to import
// fails the load (the reference renders it as an invisible empty BVH).
#include <fstream>
#include <iostream>

#include "classes.h"
#include "yaml_lite.h"

namespace hrthost {

using yamllite::Node;
using yamllite::ParseError;

namespace {

// getProperty<T> (scene.cpp:13-22)
const Node& required(const std::string& name, const Node& node) {
    const Node& p = node[name];
    if (!p) throw ParseError(node.line, "Could not find required property: " + name);
    return p;
}
int getInt(const std::string& n, const Node& node) { return required(n, node).asInt(); }
float getFloat(const std::string& n, const Node& node) { return required(n, node).asFloat(); }
std::string getString(const std::string& n, const Node& node) { return required(n, node).asString(); }
// getProperty<glm::vec3> (scene.cpp:24-47)
vec3 getVec3(const std::string& name, const Node& node) {
    const Node& p = required(name, node);
    if (!p.IsSequence()) throw ParseError(p.line, "Invalid value for vector 3: " + name);
    std::vector<float> vi = p.asFloatVector();
    if (vi.size() != 3) throw ParseError(p.line, "Invalid size for vector 3: " + name);
    return vec3(vi[0], vi[1], vi[2]);
}
// getProperty<glm::vec2> (scene.cpp:49-71)
vec2 getVec2(const std::string& name, const Node& node) {
    const Node& p = required(name, node);
    if (!p.IsSequence()) throw ParseError(p.line, "Invalid value for vector 2: " + name);
    std::vector<float> vi = p.asFloatVector();
    if (vi.size() != 2) throw ParseError(p.line, "Invalid size for vector 2: " + name);
    return vec2(vi[0], vi[1]);
}
bool fileExists(const std::string& p) { std::ifstream f(p); return (bool)f; }

}  // namespace

std::string Scene::resolve(const std::string& p) const {
    if (p.empty() || p[0] == '/' || fileExists(p)) return p;  // the reference: cwd only (scene.cpp:294-296, mesh.cpp:56)
    if (!assetDir.empty()) {
        std::string q = assetDir + "/" + p;
        if (fileExists(q)) return q;
    }
    return p;
}

void Scene::setFilmSize(int w, int h, int samples) {
    film->resize(w, h, samples);
    camera = Camera(camDesc.position, camDesc.lookAt, camDesc.up, camDesc.fov, film->getAspectRatio(), camDesc.aperture, camDesc.focus);
}

int Scene::loadScene(std::string path, std::string assetDirArg) {
    objects.clear();
    materials.clear();
    textures.clear();
    assetDir = assetDirArg;
    lastError.clear();

    // getProperty<MatVec3> (scene.cpp:73-97)
    auto getMatVec3 = [&](const std::string& name, const Node& node) -> MatVec3 {
        const Node& p = required(name, node);
        if (p.IsSequence()) return MatVec3(getVec3(name, node));
        std::string textureName = getString(name, node);
        if (textures.count(textureName) == 1) return MatVec3(textures[textureName]);
        textures[textureName] = std::make_shared<ImageTexture>(resolve(textureName));
        return MatVec3(textures[textureName]);
    };
    // getProperty<MatScalar> (scene.cpp:99-125)
    auto getMatScalar = [&](const std::string& name, const Node& node) -> MatScalar {
        const Node& p = required(name, node);
        float fv;
        if (p.tryFloat(fv)) return MatScalar(fv);
        std::string textureName = getString(name, node);
        if (textures.count(textureName) == 1) return MatScalar(textures[textureName]);
        textures[textureName] = std::make_shared<ImageTexture>(resolve(textureName));
        return MatScalar(textures[textureName]);
    };

    try {
        Node root = yamllite::LoadFile(path);
        std::cout << "Loading scene: " << path << std::endl;

        if (const Node& filmNode = root["film"]) {  // scene.cpp:140-154
            int w = getInt("width", filmNode);
            int h = getInt("height", filmNode);
            int samples = getInt("samples", filmNode);
            std::string outputPath = getString("output", filmNode);
            if (w < 2 || h < 2 || samples < 1) throw ParseError(filmNode.line, "film needs width, height >= 2 and samples >= 1");
            // what the device path takes (hrt_hip.hip check_params): refused here, before a film of that size is allocated
            if ((long long)w * (long long)h > (1ll << 30)) throw ParseError(filmNode.line, "film larger than 2^30 pixels");
            film = std::make_shared<Film>(w, h, samples, outputPath);
        } else {
            std::cout << "Must specify film descriptor!" << std::endl;
            lastError = "Must specify film descriptor!";
            return -1;
        }

        if (const Node& cameraNode = root["camera"]) {  // scene.cpp:156-172
            camDesc.position = getVec3("position", cameraNode);
            camDesc.lookAt = getVec3("look_at", cameraNode);
            camDesc.up = getVec3("up", cameraNode);
            camDesc.fov = getFloat("fov", cameraNode);
            camDesc.aperture = getFloat("aperture", cameraNode);
            camDesc.focus = getFloat("focal_distance", cameraNode);
            camera = Camera(camDesc.position, camDesc.lookAt, camDesc.up, camDesc.fov, film->getAspectRatio(), camDesc.aperture, camDesc.focus);
        } else {
            std::cout << "Must specify camera descriptor!" << std::endl;
            lastError = "Must specify camera descriptor!";
            return -1;
        }

        if (const Node& texturesNode = root["textures"]) {  // scene.cpp:174-213
            for (const Node& texture : texturesNode.seq) {
                std::string name = getString("name", texture);
                if (textures.count(name) > 0) throw ParseError(texture.line, "Texture name already exists!");
                std::string type = getString("type", texture);
                if (type == "solid") textures[name] = std::make_shared<SolidColourTexture>(getVec3("colour", texture));
                if (type == "image") textures[name] = std::make_shared<ImageTexture>(resolve(getString("path", texture)));
                if (type == "checkered") textures[name] = std::make_shared<CheckeredTexture>(getVec3("even", texture), getVec3("odd", texture));
                if (type == "environment") textures[name] = std::make_shared<EnvironmentMap>(resolve(getString("path", texture)));
            }
        }

        if (const Node& bg = root["camera"]["background"]) {  // scene.cpp:215-237
            if (bg.IsSequence()) {
                background = std::make_shared<SolidColourTexture>(getVec3("background", root["camera"]));
            } else {
                std::string textureName = getString("background", root["camera"]);
                if (textures.count(textureName) == 1) background = textures[textureName];
                else { textures[textureName] = std::make_shared<EnvironmentMap>(resolve(textureName)); background = textures[textureName]; }
            }
        } else {
            throw ParseError(root["camera"].line, "Could not find required property: background");
        }

        if (const Node& materialsNode = root["materials"]) {  // scene.cpp:239-270
            for (const Node& material : materialsNode.seq) {
                std::string name = getString("name", material);
                std::string type = getString("type", material);
                // additive types first: they do not need `albedo`
                if (type == "dielectric") {
                    MatScalar ior = getMatScalar("ior", material);
                    MatScalar rough = material["roughness"] ? getMatScalar("roughness", material) : MatScalar(0.0f);
                    materials[name] = std::make_shared<Dielectric>(ior, rough);
                    continue;
                }
                if (type == "uv_test") { materials[name] = std::make_shared<UVTest>(); continue; }
                MatVec3 albedo = getMatVec3("albedo", material);  // required for every reference type (scene.cpp:244, Q-13)
                if (type == "diffuse_light") { materials[name] = std::make_shared<DiffuseLight>(albedo, getMatScalar("strength", material)); continue; }
                if (type == "lambertian") { materials[name] = std::make_shared<Lambertian>(albedo); continue; }
                if (type == "metal") { materials[name] = std::make_shared<Metal>(albedo, getMatScalar("roughness", material)); continue; }
                if (type == "isotropic") { materials[name] = std::make_shared<Isotropic>(getVec3("albedo", material)); continue; }
                if (type == "pbr") {
                    materials[name] = std::make_shared<PBR>(getVec3("albedo", material), getFloat("metalness", material), getFloat("roughness", material));
                    continue;
                }
                // unknown material types are silently dropped (scene.cpp:246-265)
            }
        } else {
            std::cout << "Couldn't find any material descriptors!" << std::endl;
        }

        if (const Node& objectsNode = root["objects"]) {  // scene.cpp:272-363
            if (objectsNode.IsSequence()) {
                for (const Node& object : objectsNode.seq) {
                    std::shared_ptr<Hittable> o;
                    std::string type = getString("type", object);
                    std::shared_ptr<Material> m;
                    if (type != "constant_medium") {
                        std::string materialKey = getString("material", object);
                        if (materials.count(materialKey) == 1) m = materials[materialKey];
                        else { std::cout << "Material " << materialKey << " does not exist!" << std::endl; continue; }
                    }
                    auto makeBox = [&](const Node& n, std::shared_ptr<Material> mat) -> std::shared_ptr<Box> {
                        if (n["min"]) return Box::minMaxBox(getVec3("min", n), getVec3("max", n), mat);
                        return std::make_shared<Box>(getVec3("center", n), getVec3("dimensions", n), mat);
                    };
                    if (type == "mesh") {
                        std::string p = resolve(getString("path", object));
                        auto mesh = std::make_shared<Mesh>(p, m);
                        if (!mesh->loaded()) { lastError = "could not import mesh: " + p; std::cout << lastError << std::endl; return -1; }
                        o = mesh;
                    }
                    if (type == "sphere") o = std::make_shared<Sphere>(getVec3("center", object), getFloat("radius", object), m);
                    if (type == "yz_rect") { vec2 Y = getVec2("y", object), Z = getVec2("z", object); o = std::make_shared<YZRect>(Y.x, Y.y, Z.x, Z.y, getFloat("k", object), m); }
                    if (type == "xz_rect") { vec2 X = getVec2("x", object), Z = getVec2("z", object); o = std::make_shared<XZRect>(X.x, X.y, Z.x, Z.y, getFloat("k", object), m); }
                    if (type == "xy_rect") { vec2 X = getVec2("x", object), Y = getVec2("y", object); o = std::make_shared<XYRect>(X.x, X.y, Y.x, Y.y, getFloat("k", object), m); }
                    if (type == "box") o = makeBox(object, m);
                    if (type == "triangle") o = std::make_shared<Triangle>(getVec3("v0", object), getVec3("v1", object), getVec3("v2", object), m);
                    if (type == "constant_medium") {
                        const Node& b = required("boundary", object);
                        std::string bt = getString("type", b);
                        std::shared_ptr<Hittable> boundary;
                        if (bt == "sphere") boundary = std::make_shared<Sphere>(getVec3("center", b), getFloat("radius", b), nullptr);
                        else if (bt == "box") boundary = makeBox(b, nullptr);
                        else throw ParseError(b.line, "constant_medium boundary must be a sphere or a box");
                        o = std::make_shared<ConstantMedium>(boundary, getFloat("density", object), getVec3("colour", object));
                    }
                    if (!o) throw ParseError(object.line, "Unknown object type: " + type);

                    // HANDLE TRANSFORMATIONS (scene.cpp:334-354): rotate, then scale, then translate,
                    // whatever the key order in the file
                    if (const Node& transformNode = object["transform"]) {
                        if (transformNode["rotate_y"]) o = std::make_shared<RotateY>(o, getFloat("rotate_y", transformNode));
                        if (transformNode["rotate"]) {
                            vec3 angles = getVec3("rotate", transformNode);
                            vec3 rad(hrt::gradians(angles.x), hrt::gradians(angles.y), hrt::gradians(angles.z));
                            o = std::make_shared<RotateQuat>(o, hrt::quat_from_euler(rad));
                        }
                        if (transformNode["scale"]) o = std::make_shared<Scale>(o, getVec3("scale", transformNode));
                        if (transformNode["translate"]) o = std::make_shared<Translate>(o, getVec3("translate", transformNode));
                    }
                    objects.add(o);
                }
            }
        } else {
            std::cout << "Couldn't find any object descriptors!" << std::endl;
        }
    } catch (const ParseError& ex) {
        std::cout << ex.what() << std::endl;
        lastError = ex.what();
        return -1;
    } catch (const FlattenError& ex) {
        std::cout << ex.what() << std::endl;
        lastError = ex.what();
        return -1;
    }
    isLoaded = true;
    return 1;
}

}  // namespace hrthost
