/* hrt_host.h — C ABI of libhrt_host.so: the host plumbing the north star keeps
 * (YAML scene loader, mesh import seam, class surface -> flat scene, Film
 * writers) exposed to non-C++ callers (the Python test/bench harness).
 *
 * Reference anchors: Scene::loadScene (scene.cpp:127-374), Scene getters
 * (scene.h:29-33), Mesh::Mesh (mesh.cpp:13-41), Film::outputFilm
 * (film.cpp:59-79).  Nothing here touches the GPU; libhrt_hip.so (hrt.h)
 * consumes the `hrt_flat_scene` this library produces.
 */
#ifndef HRT_HOST_H
#define HRT_HOST_H

#include "hrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hrt_host_scene hrt_host_scene;

/* Scene::loadScene.  `asset_dir` (may be NULL) is searched for relative mesh /
 * texture paths that do not exist relative to the cwd.  On failure returns
 * HRT_ERR_PARSE / HRT_ERR_IO and hrt_host_last_error() holds the loader's message. */
hrt_status hrt_host_load_yaml(const char* yaml_path, const char* asset_dir, hrt_host_scene** out);
void hrt_host_free(hrt_host_scene* s);

/* The flattened world (getScene() + getBackground()); valid until hrt_host_free. */
const hrt_flat_scene* hrt_host_flat(const hrt_host_scene* s);
/* film: {width, height, samples, output} of the YAML file (scene.cpp:140-149). */
hrt_status hrt_host_film(const hrt_host_scene* s, int32_t* width, int32_t* height, int32_t* samples, char* output,
                         int32_t output_cap);
/* Camera constants for a film of width x height (aspect = width / height,
 * scene.cpp:165); pass the YAML film size for the file's own camera. */
hrt_status hrt_host_camera(const hrt_host_scene* s, int32_t width, int32_t height, hrt_camera* out);
/* Depth (nodes on the longest root-to-leaf path) of mesh `mesh`'s flattened BVH. */
int32_t hrt_host_bvh_depth(const hrt_host_scene* s, int32_t mesh);

/* Who builds the culling tree of a mesh (the BVHNode constructor's job, bvh.cpp:6-61).  Default (fn == NULL): the host's binned-SAH
 * builder.  fn = hrt_bvh_build_device of libhrt_hip.so (include/hrt.h; this library does not link it): meshes of more than two
 * triangles are built on GPU `device` as a Morton-ordered LBVH -- about a fifth more node visits per ray, built in milliseconds.
 * Affects scenes loaded afterwards; a failing device build fails the load (no silent fall-back to the host builder). */
typedef hrt_status (*hrt_host_bvh_build_fn)(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                                            uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out);
void hrt_host_set_bvh_builder(hrt_host_bvh_build_fn fn, int device);

/* hrt_params with the reference's constants: MAX_DEPTH 50, t_min 0.001,
 * quirks = HRT_QUIRKS_REFERENCE, seed 0. */
void hrt_default_params(hrt_params* p, int32_t width, int32_t height, int32_t samples);

/* Procedural stand-ins for the assets the reference's scenes name but do not
 * ship.  The mesh writers return the triangle count (< 0 on error). */
int64_t hrt_asset_write_teapot_obj(const char* path, double detail);
int64_t hrt_asset_write_bust_obj(const char* path, double detail);
hrt_status hrt_asset_write_hall_hdr(const char* path, int32_t width, int32_t height);

/* Film::outputFilm by suffix (.png / .tga / else BMP), u8 RGB rows top first. */
hrt_status hrt_host_write_image(const char* path, const uint8_t* rgb, int32_t width, int32_t height);
/* Codecs, for tests: Radiance .hdr -> 3 x fp32 (caller buffer of w*h*3 floats; call
 * with out == NULL to query the size), PNG -> 3 x u8. */
hrt_status hrt_host_read_hdr(const char* path, int32_t* width, int32_t* height, float* out, int64_t out_cap_floats);
hrt_status hrt_host_read_png(const char* path, int32_t* width, int32_t* height, uint8_t* out, int64_t out_cap_bytes);
/* Baseline JPEG -> 3 x u8 (what ImageTexture gets from stbi_load(path, ..., 3), texture.cpp:34-36). */
hrt_status hrt_host_read_jpeg(const char* path, int32_t* width, int32_t* height, uint8_t* out, int64_t out_cap_bytes);
hrt_status hrt_host_write_hdr(const char* path, const float* rgb, int32_t width, int32_t height);
/* The fp32 linear film bit for bit (Portable Float Map, little-endian, colour): the dump to diff two renders exactly
 * (Radiance RGBE keeps 8 mantissa bits).  Rows are handed over top first, like everywhere else. */
hrt_status hrt_host_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height);
hrt_status hrt_host_read_pfm(const char* path, int32_t* width, int32_t* height, float* out, int64_t out_cap_floats);

const char* hrt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HRT_HOST_H */
