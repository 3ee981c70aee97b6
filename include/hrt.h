/* hrt.h — C ABI of libhrt_hip.so: the MI355X (gfx950) replacement for the
 * reference's per-pixel Monte-Carlo render loop.
 *
 * The reference (Todegal/HobbyRaytracer) has no plugin/FFI interface; its only
 * seam for this path is the free function
 *     static void render(int nThreads, const std::shared_ptr<Texture> background,
 *                        const std::shared_ptr<Hittable> world, const Camera& camera,
 *                        std::shared_ptr<Film>& film)          (main.cpp:81-82)
 * called once from main() (main.cpp:176) on objects produced by the Scene
 * getters (main.cpp:158-162, scene.h:29-33).  Everything that function does
 * between "object graph" and "Film::pixels" is what this library replaces:
 *     rayColour            main.cpp:38-79
 *     render pixel loop    main.cpp:111-135
 *     Hittable::hit tree   hittableList.cpp:4-21, bvh.cpp:69-78, triangle.cpp:57-131,
 *                          sphere.cpp:20-49, aarect.h:12-39/59-86/106-133, box.h:27-55,
 *                          translate.cpp:7-19, scale.cpp:11-27, rotateQuat.cpp:44-66,
 *                          rotateY.cpp:44-75, constantMedium.cpp:4-38
 *     Material::scatter    material.h:79-85,96-104,116-129,137-153,166-177,204-229, material.cpp:18-28
 *     Texture::colourValue texture.cpp:17-28,53-74,76-97
 *     Film::tonemap/writeColour  film.cpp:25-52
 * INTEGRATION.md shows the ~40-line binding a maintainer of the reference adds.
 *
 * Conventions: plain C structs, pointers and sizes only; the library copies
 * what it is given at hrt_scene_create (caller keeps ownership of host arrays);
 * every entry point returns hrt_status (0 = ok) and never throws;
 * hrt_last_error() returns the HIP error text of the calling thread's last
 * failure.  All arithmetic is fp32 except Dielectric's Fresnel term (fp64,
 * material.h:210-218).
 */
#ifndef HRT_H
#define HRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hrt_status {
    HRT_OK = 0,
    HRT_ERR_INVALID = 1,      /* bad argument / inconsistent flat scene */
    HRT_ERR_HIP = 2,          /* a HIP runtime call failed: see hrt_last_error() */
    HRT_ERR_NO_DEVICE = 3,    /* no gfx950 device visible */
    HRT_ERR_OOM = 4,
    HRT_ERR_IO = 5,
    HRT_ERR_PARSE = 6,
    HRT_ERR_UNSUPPORTED = 7
} hrt_status;

/* ---- flattened scene (what Hittable::flatten() emits) ------------------- */

/* Top-level object kinds, in the reference's class vocabulary. */
enum {
    HRT_PRIM_SPHERE = 0,   /* sphere.cpp      p = cx,cy,cz,r                     */
    HRT_PRIM_XY_RECT = 1,  /* aarect.h:106    p = x0,x1,y0,y1,k                  */
    HRT_PRIM_XZ_RECT = 2,  /* aarect.h:59     p = x0,x1,z0,z1,k                  */
    HRT_PRIM_YZ_RECT = 3,  /* aarect.h:12     p = y0,y1,z0,z1,k                  */
    HRT_PRIM_BOX = 4,      /* box.h           p = min.xyz, max.xyz               */
    HRT_PRIM_MESH = 5,     /* mesh.cpp        mesh = index into meshes[]         */
    HRT_PRIM_MEDIUM = 6,   /* constantMedium.cpp  boundary_kind + p, density     */
    HRT_PRIM_TRIANGLE = 7  /* triangle.cpp:4-40  Triangle (NOT the mesh's ITriangle: nothing in the reference
                              constructs one)   p = v0.xyz, v1.xyz, v2.xyz */
};

/* Instance wrappers (translate.cpp, scale.cpp, rotateQuat.cpp, rotateY.cpp).
 * The chain is stored OUTERMOST FIRST, i.e. in the order the ray meets them. */
enum {
    HRT_XF_TRANSLATE = 0,  /* v = offset.xyz                                     */
    HRT_XF_SCALE = 1,      /* v = factor.xyz                                     */
    HRT_XF_ROTATE_QUAT = 2,/* v = quat x,y,z,w                                   */
    HRT_XF_ROTATE_Y = 3    /* v = sinTheta, cosTheta                             */
};
#define HRT_MAX_XFORMS 4

typedef struct hrt_xform {
    int32_t kind;
    float v[4];
} hrt_xform;

typedef struct hrt_prim {
    int32_t kind;
    int32_t material;        /* index into materials[] (MEDIUM: the Isotropic phase function) */
    int32_t mesh;            /* HRT_PRIM_MESH only */
    int32_t boundary_kind;   /* HRT_PRIM_MEDIUM only: HRT_PRIM_SPHERE or HRT_PRIM_BOX */
    float p[9];
    float density;           /* HRT_PRIM_MEDIUM only */
    int32_t n_xforms;
    hrt_xform xf[HRT_MAX_XFORMS];
} hrt_prim;

enum {
    HRT_MAT_LAMBERTIAN = 0,    /* material.h:132-157  albedo                     */
    HRT_MAT_METAL = 1,         /* material.h:159-182  albedo, s0 = roughness     */
    HRT_MAT_DIELECTRIC = 2,    /* material.h:199-242  s0 = ir, s1 = roughness    */
    HRT_MAT_DIFFUSE_LIGHT = 3, /* material.h:91-109   albedo = emit, s0 = strength */
    HRT_MAT_ISOTROPIC = 4,     /* material.h:73-89    albedo                     */
    HRT_MAT_PBR = 5,           /* material.cpp:4-28   albedo, s0 = roughness, mix_tex */
    HRT_MAT_UVTEST = 6         /* material.h:111-130                             */
};

typedef struct hrt_matvec3 {   /* material.h:10-35 MatVec3: constant or texture */
    int32_t tex;               /* < 0 : constant c */
    float c[3];
} hrt_matvec3;

typedef struct hrt_matscalar { /* material.h:37-58 MatScalar: constant or length(texture rgb) */
    int32_t tex;
    float c;
} hrt_matscalar;

typedef struct hrt_material {
    int32_t kind;
    hrt_matvec3 albedo;
    hrt_matscalar s0;
    hrt_matscalar s1;
    int32_t mix_tex;
} hrt_material;

enum {
    HRT_TEX_SOLID = 0,    /* texture.h:18-21   c                                  */
    HRT_TEX_CHECKER = 1,  /* texture.cpp:17-28 even, odd = texture indices        */
    HRT_TEX_IMAGE = 2,    /* texture.cpp:53-74 u8 RGB, texels_u8 + offset         */
    HRT_TEX_ENV = 3       /* texture.cpp:76-97 fp32, `channels` per texel, texels_f32 + offset */
};

typedef struct hrt_texture {
    int32_t kind;
    float c[3];
    int32_t even, odd;
    int32_t width, height, channels;
    int32_t _pad;
    uint64_t offset;      /* element offset into texels_u8 / texels_f32; width==0 => "no data" (cyan) */
} hrt_texture;

/* One triangle mesh = a contiguous range of the triangle arrays plus its
 * flattened BVH (node indices are relative to node_first). */
typedef struct hrt_mesh {
    uint32_t tri_first, tri_count;
    uint32_t node_first, node_count;
} hrt_mesh;

/* 64-byte BVH node holding the boxes of BOTH children, so one fetch tests
 * two boxes (32 B per box, the unit SURVEY.md §8(d) prices a "node visit" at).
 *   child >= 0 : index of an inner node (relative to the mesh's node_first)
 *   child <  0 : leaf, ~child = (first_tri_in_mesh << 3) | (tri_count - 1)
 * An unused child slot has an inverted box (min = +inf, max = -inf). */
typedef struct hrt_bvh_node {
    float c0_min_x, c0_max_x, c0_min_y, c0_max_y;
    float c1_min_x, c1_max_x, c1_min_y, c1_max_y;
    float c0_min_z, c0_max_z, c1_min_z, c1_max_z;
    int32_t child0, child1;
    int32_t _pad0, _pad1;
} hrt_bvh_node;

typedef struct hrt_flat_scene {
    uint32_t n_prims;      const hrt_prim* prims;          /* world list order (hittableList.cpp:12) */
    uint32_t n_materials;  const hrt_material* materials;
    uint32_t n_textures;   const hrt_texture* textures;
    uint32_t n_meshes;     const hrt_mesh* meshes;
    uint64_t n_tris;
    const float* tri_pos;  /* 9 floats per triangle: v0.xyz v1.xyz v2.xyz  (triangle.h:31) */
    const float* tri_nrm;  /* 9 floats per triangle */
    const float* tri_uv;   /* 6 floats per triangle */
    const float* tri_box;  /* 6 floats per triangle (min.xyz, max.xyz): the box of the LOWEST BVHNode that
                              holds the triangle in the reference's own tree (bvh.cpp:20-36,52-60 over the
                              padded ITriangle boxes of triangle.cpp:133-151).  The reference rejects a
                              triangle hit whose leaf-level box fails AABB::hit (bvh.cpp:71), which matters
                              for the t < t_min self-hits of Q-2; the flattened BVH applies the same test to
                              accepted candidates so results do not depend on ITS topology.  NULL = derived by the
                              library from tri_pos and tri_ref_order (hrt_pack.h pack_ref_tree). */
    const uint32_t* tri_ref_order; /* per triangle: (node << 1) | side, where `node` numbers the lowest
                              BVHNodes of the reference's own tree for the mesh in depth-first order and
                              `side` is 0 for that node's `left` child, 1 for `right` (bvh.cpp:20-36).  Once
                              BVHNode::hit has accepted a hit with t < t_min every later BOX test fails
                              (bvh.cpp:71 with t_max = rec.t), so among several such self-hits the reference
                              keeps the one in the FIRST node its walk meets — except that the `right`
                              triangle of that same node is still tested (no box in between, bvh.cpp:75) and
                              wins if it is not farther.  The flattened traversal reproduces exactly that.
                              The codes also DEFINE the reference's whole tree for the library: the reference sorts and
                              splits at start + n / 2 (bvh.cpp:39-43), so a node is a contiguous range of the depth-first
                              order and its box the union of the padded triangle boxes of the range; rays whose
                              ITriangle::hit arithmetic has gone meaningless (quirk Q-4 with a vanishing direction
                              component on the shear axis) are walked through THAT tree, node by node in its order
                              (hrt_device.h ref_walk).  hrt_scene_create refuses codes that are not the depth-first
                              code of such a tree.  NULL = the triangles in the given order (a tree whose sorts
                              changed nothing). */
    uint64_t n_nodes;      const hrt_bvh_node* nodes;
    uint64_t n_texels_u8;  const uint8_t* texels_u8;
    uint64_t n_texels_f32; const float* texels_f32;
    int32_t background_tex;  /* main.cpp:58 background->colourValue(u, v, 0) */
    int32_t _pad;
} hrt_flat_scene;

/* camera.h:41-48 — the constants of Camera::getRay.  The reference hard-wires the lens offset to 0 (camera.h:34:
 * `rd = {0,0,0}; // glm::circularRand(lensRadius)`, "TODO: Add back in randomness"), so lens_u / lens_v / lens_radius are only
 * read with HRT_FLAG_THIN_LENS, which puts that commented-out call back: rd = circularRand(lensRadius) -- a point ON the circle
 * of that radius, as glm defines it --, offset = u * rd.x + v * rd.y (camera.h:35-37). */
typedef struct hrt_camera {
    float origin[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float lens_u[3];        /* camera.h:20  u = normalize(cross(up, w)) */
    float lens_v[3];        /* camera.h:21  v = cross(w, u)             */
    float lens_radius;      /* camera.h:26  aperture / 2                */
} hrt_camera;

/* Quirk switches.  A set bit = reference behaviour. */
enum {
    HRT_Q1_ROTQ_NORMALIZE = 1u << 0,  /* rotateQuat.cpp:51 normalises the direction, t units change */
    HRT_Q2_TRI_NO_TMIN = 1u << 1,     /* triangle.cpp:106-109 has no t_min test                    */
    HRT_Q3_TRI_NO_FACE = 1u << 2,     /* triangle.cpp:118-128 never calls setFaceNormal            */
    HRT_Q4_SHEAR_FROM_ORIGIN = 1u << 3/* triangle.cpp:70 picks kZ from the ray ORIGIN              */
};
#define HRT_QUIRKS_REFERENCE 0xFu
#define HRT_QUIRKS_FIXED 0x0u

typedef struct hrt_params {
    int32_t width, height;   /* film_desc.dimensions (film.h:3-7) */
    int32_t samples;         /* film_desc.samples                  */
    int32_t max_depth;       /* MAX_DEPTH = 50 (main.cpp:32)       */
    float t_min;             /* 0.001f (main.cpp:45)               */
    uint32_t quirks;
    uint32_t seed_lo, seed_hi;
    uint32_t flags;          /* HRT_FLAG_* */
} hrt_params;

/* hrt_params.flags */
enum {
    HRT_FLAG_STATS = 1u << 0,  /* also count box_tests / tri_tests / mesh_hits / env_lookups (the counting
                                  build of the kernels; rays and samples are always counted) */
    HRT_FLAG_MEGAKERNEL = 1u << 1, /* render with the single persistent-lanes kernel (k_pathtrace) instead of the
                                  default wavefront pipeline (k_wf_*); results are bit-identical */
    HRT_FLAG_TIMING = 1u << 2, /* wavefront pipeline: also time the traversal kernel's launches with HIP events
                                  (hrt_stats.traversal_ms) */
    HRT_FLAG_THIN_LENS = 1u << 3, /* sample the lens as camera.h:34's commented-out call would (see hrt_camera); off = the reference */
    HRT_FLAG_PROGRESS = 1u << 4 /* keep the host-readable progress counter of hrt_scene_progress / hrt_multi_progress up to date
                                    (the reference's reporter thread, main.cpp:97-109): one tiny launch per round */
};

typedef struct hrt_rect { int32_t x0, y0, w, h; } hrt_rect;   /* y0 = row index from the TOP (pIdx / W) */

typedef struct hrt_stats {
    uint64_t rays;       /* path segments = iterations of main.cpp:43-45 */
    uint64_t samples;    /* camera samples                                */
    uint64_t box_tests;  /* BVH child boxes tested (32 B each)            */
    uint64_t tri_tests;  /* triangles tested (36 B each)                  */
    uint64_t mesh_hits;  /* segments whose closest hit is a mesh triangle (60 B attrs) */
    uint64_t env_lookups;/* segments that escaped to an fp32 env map (12 B) */
    double kernel_ms;    /* path-trace time (megakernel launch, or the whole wavefront pipeline of one render
                            call), summed over `launches`, from HIP events recorded on the launch stream */
    uint64_t launches;   /* render calls (megakernel launches / wavefront pipeline runs) accumulated here */
    double traversal_ms; /* wavefront pipeline with HRT_FLAG_TIMING: time of the BVH traversal kernel (k_wf_ext),
                            summed over its `traversal_launches` launches, from HIP events around each launch */
    uint64_t traversal_launches;
    uint64_t traversal_box_tests, traversal_tri_tests;   /* the part of box_tests / tri_tests counted inside those k_wf_ext
                            launches (HRT_FLAG_STATS): rounds that run inside the task-persistent tail kernel are not in it */
} hrt_stats;

typedef struct hrt_hit {          /* hitRecord (hittable.h:8-25) as seen by rayColour */
    float t;
    int32_t prim;                 /* -1 = miss */
    int32_t tri;                  /* triangle index within the mesh, -1 otherwise */
    int32_t front_face;           /* hittable.h:19.  Under quirk Q-3 a mesh hit never writes it (triangle.cpp:118-128): for a mesh
                                     that stands in the world list without a wrapper it is the flag of the previous successful object
                                     of HittableList::hit's walk (hittableList.cpp:6-16), `1` when there was none */
    float p[3];
    float normal[3];
    float u, v;
} hrt_hit;

/* The culling tree of one mesh, built on `device`: a Morton-ordered LBVH (csrc/hrt_lbvh.hip).  Stands where the reference has
 * the BVHNode constructor (bvh.cpp:6-61) -- for the TOPOLOGY only, like the host's binned-SAH builder (host/bvh_build.cpp): the
 * closest hit does not depend on it.  tri_pos: 9 floats per triangle (host memory), finite; max_leaf in 1..8, n_tris > max_leaf.
 * Out (host memory, caller-allocated): nodes_out[n_tris - 1] (hrt_bvh_node, root = 0, child boxes = padded ITriangle boxes
 * (triangle.cpp:133-151) united and widened by the kernels' rounding guard), *n_nodes_out, order_out[n_tris] = the triangle at each
 * position of the leaf order the nodes' leaf codes refer to, *depth_out = inner-node levels (the traversal's stack need). */
hrt_status hrt_bvh_build_device(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                                uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out);
/* The same interface, THE host builder's tree: host/bvh_build.cpp's binned-SAH algorithm run on the device with the same decisions
 * and arithmetic (csrc/hrt_sahbvh.hip) -- the same topology, up to the order of the triangles inside a leaf.  HRT_ERR_UNSUPPORTED
 * when a large node needs the host's median split (exhausted depth budget): the caller builds that mesh on the host. */
hrt_status hrt_bvh_build_sah(int device, const float* tri_pos, uint32_t n_tris, uint32_t max_leaf, hrt_bvh_node* nodes_out,
                             uint32_t* n_nodes_out, uint32_t* order_out, int32_t* depth_out);

typedef struct hrt_scene hrt_scene;   /* device-resident flattened scene */

hrt_status hrt_device_count(int* n);

/* Uploads (copies) the flat scene to `device`.  Validates every index. */
hrt_status hrt_scene_create(const hrt_flat_scene* flat, int device, hrt_scene** out);
void hrt_scene_destroy(hrt_scene* scene);

/* Blocking: renders tile (x0,y0,w,h) of the W x H film and writes
 * w*h*3 fp32 LINEAR radiance means (row-major within the tile, row 0 = top)
 * to the caller-owned HOST buffer.  This is render() of main.cpp:81-140 up to
 * and including `pixelColour /= samples` (main.cpp:126). */
hrt_status hrt_render_tile(hrt_scene* scene, const hrt_camera* cam, const hrt_params* params, hrt_rect tile,
                           float* out_rgb_linear, hrt_stats* stats);

/* Asynchronous multi-GPU form: renders the interleaved row blocks owned by
 * `rank` of `n_ranks` (block b = rows [b*rows_per_block, (b+1)*rows_per_block)
 * belongs to rank b % n_ranks) into a DEVICE buffer of
 * hrt_stripe_rows(height, rows_per_block, rank, n_ranks) * W * 3 floats, rows in
 * increasing absolute row order, on HIP stream `stream` (NULL = default
 * stream).  Counters are accumulated on the device; fetch them with
 * hrt_scene_stats() after synchronising the stream. */
hrt_status hrt_render_stripes_device(hrt_scene* scene, const hrt_camera* cam, const hrt_params* params,
                                     int32_t rows_per_block, int32_t rank, int32_t n_ranks, float* d_out_rgb_linear,
                                     void* stream);
/* Progressive / resumable form (the reference's render() takes all samples of a pixel in one go,
 * main.cpp:118-126).  Adds samples [sample_first, sample_first + sample_count) of params->samples to the
 * accumulation buffer (same layout as hrt_render_stripes_device's output): it holds the running SUM of the samples'
 * radiances, added in sample order.  sample_first == 0 starts a new accumulation (the buffer need not be
 * cleared).  The call whose range reaches params->samples divides the sums by params->samples (main.cpp:126),
 * after which the buffer is bit-identical to a one-shot render with the same params, however the samples
 * were batched.  The buffer plus the next sample index is the whole checkpoint of a render: it can be copied out,
 * stored, and continued later (the row layout depends on rows_per_block / rank / n_ranks only).  Between passes a
 * preview is accum / samples_done.  sample_count < 0 means "all that are left" (params->samples - sample_first); a range
 * outside [0, params->samples) or an empty one is HRT_ERR_INVALID.  Wavefront pipeline only: HRT_FLAG_MEGAKERNEL with a
 * partial range is HRT_ERR_UNSUPPORTED. */
hrt_status hrt_render_stripes_accumulate_device(hrt_scene* scene, const hrt_camera* cam, const hrt_params* params,
                                                int32_t rows_per_block, int32_t rank, int32_t n_ranks, float* d_accum,
                                                int32_t sample_first, int32_t sample_count, void* stream);
/* Blocking host-buffer form: `accum` is uploaded first when sample_first > 0, and downloaded after the pass. */
hrt_status hrt_render_stripes_accumulate(hrt_scene* scene, const hrt_camera* cam, const hrt_params* params,
                                         int32_t rows_per_block, int32_t rank, int32_t n_ranks, float* accum,
                                         int32_t sample_first, int32_t sample_count, hrt_stats* stats);
/* Blocking host-buffer form of the above (used by the CLI's one-thread-per-GPU
 * scheduler): same row layout, output copied to the caller-owned HOST buffer. */
hrt_status hrt_render_stripes(hrt_scene* scene, const hrt_camera* cam, const hrt_params* params, int32_t rows_per_block,
                              int32_t rank, int32_t n_ranks, float* out_rgb_linear, hrt_stats* stats);
int32_t hrt_stripe_rows(int32_t height, int32_t rows_per_block, int32_t rank, int32_t n_ranks);
/* Absolute row index of local row `local` of rank's stripes; -1 if out of range. */
int32_t hrt_stripe_row_index(int32_t height, int32_t rows_per_block, int32_t rank, int32_t n_ranks, int32_t local);

/* ---- multi-GPU session -----------------------------------
 * The reference's render() (main.cpp:81-140) has one parallel loop over all pixels of the film (main.cpp:111-135, no state
 * shared between pixels).  Here the flattened scene is replicated on `n_devices` GPUs of THIS process (`devices` = their
 * indices, NULL = 0..n-1), the film rows are dealt to them in interleaved blocks (hrt_stripe_rows) and every device keeps
 * the running sums of its rows in its own memory.  With more than one device (or force_rccl != 0) the session owns an RCCL
 * communicator per device (ncclCommInitAll) and hrt_multi_render gathers the device-resident stripes on the first device
 * with one grouped ncclAllGather of equal, padded shares over xGMI; nothing but the finished film crosses PCIe.
 * force_rccl < 0 is LOOPBACK, a test mode for boxes with fewer devices than ranks: a device may be listed once per logical
 * rank and the gather is one device-to-device copy per rank on that rank's stream instead of ncclAllGather (RCCL refuses two
 * ranks on one device); one host thread, stream and scene per rank, the padded shares, idle ranks and the film assembly are
 * the production code.
 * If a rank fails inside hrt_multi_render the others have already added the sample range to their sums: the session then
 * refuses to continue until it is given resume_sums (a checkpoint) or started again at sample 0. */
typedef struct hrt_multi hrt_multi;
hrt_status hrt_multi_create(const hrt_flat_scene* flat, int32_t n_devices, const int32_t* devices, int32_t force_rccl, hrt_multi** out);
void hrt_multi_destroy(hrt_multi* m);
int32_t hrt_multi_devices(const hrt_multi* m);
int32_t hrt_multi_uses_rccl(const hrt_multi* m);
/* Adds samples [sample_first, sample_first + sample_count) of params->samples on all devices at once (sample_count < 0: all that
 * are left; 0: none, only gather what is there), gathers, and hands out (both optional, caller-owned HOST buffers):
 *   out_sums  W*H*3 floats in film order: the running sums -- the means once the range has reached params->samples
 *             (main.cpp:126); this plus the next sample index is the checkpoint of the render;
 *   out_u8    W*H*3 bytes: Film::tonemap + writeColour (film.cpp:25-52) of sums / samples_done, resolved on the first device.
 * resume_sums != NULL (W*H*3 floats, film order): the devices' sums are set from it first (continuing from a checkpoint,
 * with any device count).  Blocking.  `stats`: summed over the devices; the two times are the slowest device's. */
hrt_status hrt_multi_render(hrt_multi* m, const hrt_camera* cam, const hrt_params* params, int32_t rows_per_block, int32_t sample_first,
                            int32_t sample_count, const float* resume_sums, float* out_sums, uint8_t* out_u8, hrt_stats* stats);

/* The reference's progress counter (main.cpp:95-109: `pixelsCompleted`, printed every 500 ms by a reporter thread) for a path
 * that finishes its pixels together: *paths_done = camera paths (pixel samples) finished so far in the render call that is
 * running -- or ran last -- on `scene` with HRT_FLAG_PROGRESS, *paths_total = all paths of that call.  The counter lives in
 * host-mapped memory the device writes after every round: these calls read it without any HIP call or lock, from any thread,
 * while the render runs.  hrt_multi_progress sums the session's ranks.  (pixels = width * height * done / total.) */
hrt_status hrt_scene_progress(const hrt_scene* scene, uint64_t* paths_done, uint64_t* paths_total);
hrt_status hrt_multi_progress(const hrt_multi* m, uint64_t* paths_done, uint64_t* paths_total);

/* Reads and clears the device-side counters of `scene` (synchronises its device). */
hrt_status hrt_scene_stats(hrt_scene* scene, hrt_stats* stats);

/* Film::tonemap + Film::writeColour (film.cpp:25-52) on the GPU.
 * Host-buffer form and device-buffer form. */
hrt_status hrt_resolve_u8(hrt_scene* scene, const float* rgb_linear, int64_t n_pixels, uint8_t* out_rgb8);
hrt_status hrt_resolve_u8_device(hrt_scene* scene, const float* d_rgb_linear, int64_t n_pixels, uint8_t* d_out_rgb8,
                                 void* stream);

/* Test entry: world->hit(r, t_min, t_max, rec) of
 * main.cpp:45 for n rays given as host arrays o[3n], d[3n].  `pixel0` keys the
 * RNG used by ConstantMedium::hit: ray i draws as (pixel0 + i, sample 0, bounce 0). */
hrt_status hrt_closest_hit(hrt_scene* scene, const hrt_params* params, int64_t n, const float* o, const float* d,
                           float t_min, float t_max, uint32_t pixel0, hrt_hit* out);

/* Test entry: evaluates the shared math kernels on the GPU so that tests can
 * check CPU == GPU bit for bit.  op: 0 sin, 1 cos, 2 acos, 3 atan2(x=in, y=in2), 4 log,
 * 5 philox (in = counter words as float bits; out 4 words per input). */
hrt_status hrt_math_probe(int device, int32_t op, int64_t n, const float* in, const float* in2, float* out);

/* Debug entry: a library built with -DHRT_DEBUG_BOUNDS checks every table index a hit record is built from (triangle of a mesh,
 * prim, material, texture, mesh, frontFace source) against its table, counts the violations per kind and carries on with index 0
 * instead of faulting.  Reads and clears the 8 counters of `device`; HRT_ERR_UNSUPPORTED from a normal build
 * (tests/tools/debug_bounds.sh runs the GPU suite on such a build). */
hrt_status hrt_debug_bounds_violations(int32_t device, int64_t* out8);

const char* hrt_status_str(hrt_status s);
const char* hrt_last_error(void);
const char* hrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HRT_H */
