/*
 * rtgpu.h -- C ABI of the MI355X render path (librtgpu.so).
 *
 * This is the drop-in boundary for the reference's per-pixel render loop:
 *   raytracer/src/scene.rs:220-246   pub fn Scene::render(&self) -> Vec<Color>
 * A caller flattens its Scene (scene.rs:35-41) into the plain-data
 * rt_scene_desc below, uploads it once and asks for the frame.  Everything
 * here is POD: plain pointers and sizes, little endian, f64 geometry exactly
 * as the reference holds it, f32 only where the reference is f32.
 *
 * Conventions
 *   - every export returns RT_OK (0) or a negative RT_ERR_* code; no C++
 *     exception and no HIP error crosses this boundary;
 *   - the library never keeps a caller pointer past the call that received
 *     it (rt_scene_upload deep-copies into device memory);
 *   - output buffers are caller-owned;
 *   - one rt_ctx is used by one thread at a time; several contexts may exist.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
#define RT_OK               0
#define RT_ERR_INVALID     (-1)  /* malformed description / bad argument          */
#define RT_ERR_UNSUPPORTED (-2)  /* valid scene the device path does not cover   */
#define RT_ERR_DEVICE      (-3)  /* HIP runtime failure (no GPU, launch error..) */
#define RT_ERR_NOMEM       (-4)  /* host or device allocation failed             */

/* ---- closed sets (object/mod.rs:61-70, lighting/mod.rs:36-43,
 *      skybox.rs:8-11, material.rs:108-117) -------------------------------- */
enum { RT_OBJ_SPHERE = 0, RT_OBJ_PLANE = 1, RT_OBJ_BOX = 2, RT_OBJ_MESH = 3 };
enum { RT_TEX_SOLID = 0, RT_TEX_IMAGE = 1, RT_TEX_CHECKER = 2 };
enum { RT_LIGHT_SUN = 0, RT_LIGHT_POINT = 1, RT_LIGHT_AREA = 2 };
enum { RT_SKY_SOLID = 0, RT_SKY_NORMAL = 1, RT_SKY_CUBEMAP = 2 };
enum { RT_AREA_SPHERE = 0, RT_AREA_RECTANGLE = 1 };

#define RT_ID_NONE 0xFFFFFFFFu

/* camera.rs:5-27.  chf is the caller's precomputed field (camera.rs:45-47). */
typedef struct rt_camera {
    int32_t vw, vh;
    double  origin[3];
    double  yaw, pitch, fov, chf;
} rt_camera;

/* scene.rs:17-32 */
typedef struct rt_options {
    uint32_t max_ray_depth;
    uint8_t  ambient[3];
    uint8_t  _pad;
} rt_options;

/* material.rs:108-117.  IMAGE: rgb = w*h*3 bytes, row-major, top-left first
 * (image::RgbImage after into_rgb8, sdl/src/interpret.rs:1115). */
typedef struct rt_texture {
    uint32_t       kind;
    uint8_t        a[3];      /* SOLID colour / CHECKER colour a */
    uint8_t        b[3];      /* CHECKER colour b                */
    uint8_t        _pad[2];
    uint32_t       w, h;
    const uint8_t* rgb;
} rt_texture;

/* material.rs:147-164 */
typedef struct rt_material {
    uint32_t texture;         /* index into textures[] */
    uint32_t _pad;
    double   reflectiveness, transparency, ior, emissivity;
} rt_material;

/* object list entry, in scene order (scene.rs:36, scanned in order :61-66) */
typedef struct rt_object {
    uint32_t kind;            /* RT_OBJ_*                          */
    uint32_t material;        /* index into materials[]            */
    uint32_t payload;         /* index into the array of its kind  */
    uint32_t _pad;
} rt_object;

typedef struct rt_sphere { double c[3]; double r; } rt_sphere;            /* sphere.rs:12-21 */
typedef struct rt_plane  { double p0[3]; double n[3]; float uv_wrap; float _pad; } rt_plane; /* plane.rs:11-25 */
typedef struct rt_box    { double pos[3]; double half[3]; } rt_box;       /* aabb.rs:13-16 (size = half extent) */

/* acceleration.rs:407-417 TreeNode, flattened in depth-first pre-order
 * (a before b, the order mesh.rs:326-345 walks it).
 *   count == 0 : Branch; child a = this index + 1, child b = right_or_first
 *   count  > 0 : Leaf;   prim_indices[right_or_first .. right_or_first+count)
 * min/max = the node's `bounding` (centroid is (min+max)*0.5, :76,:116). */
typedef struct rt_bvh_node {
    double   min[3], max[3];
    uint32_t right_or_first;
    uint32_t count;
} rt_bvh_node;

/* mesh.rs:91-119.  Index triples are separate for positions, normals and
 * texture coordinates.  tri_uv may be NULL (no texcoords => uv (0,0)). */
typedef struct rt_mesh {
    const double*      verts;      uint32_t n_verts;                       /* [n][3] */
    /* Sequences (sdl/src/main.rs:70-95 rebuilds and renders the scene once per frame):
     *   0  uploaded afresh;
     *   1  this mesh -- geometry, normals, texture coordinates and tree -- is the one that stood at the same index in
     *      the previous successful rt_scene_upload of the context: its device copy is kept and the arrays here are not
     *      read (the counts still are, and must match);
     *   2  same connectivity (tri_pos, tri_nrm, tri_uv, texcoords) and same tree as that mesh, but `verts` and `normals`
     *      (same counts) are new: the kept tree's boxes and triangle records are recomputed on the device from them. */
    uint32_t           unchanged;
    const double*      normals;    uint32_t n_normals;    uint32_t _p1;   /* [n][3] */
    const float*       texcoords;  uint32_t n_texcoords;  uint32_t _p2;   /* [n][2] */
    const uint32_t*    tri_pos;                                            /* [n_tris][3] */
    const uint32_t*    tri_nrm;                                            /* [n_tris][3] */
    const uint32_t*    tri_uv;                                             /* [n_tris][3] or NULL */
    uint32_t           n_tris;     uint32_t _p3;
    /* The tree of Mesh::generate_sbvh (mesh.rs:257-272), built by the caller.  nodes == NULL with n_nodes == 0: no tree
     * is handed over and the library builds one on the device (a linear BVH over the triangles' centroids; milliseconds
     * for a million triangles, where the host's binned-SAH build takes a third of a second).  Frames do not depend
     * on which tree is used, except between two triangles hit at exactly the same distance, where the earlier one
     * in the tree's leaf order wins (mesh.rs:365-376: the earlier candidate of the reference's own tree). */
    const rt_bvh_node* nodes;      uint32_t n_nodes;      uint32_t _p4;
    const uint32_t*    prim_indices; uint32_t n_prims;    uint32_t _p5;
} rt_mesh;

/* lighting/sun.rs:13-39, point.rs:12-34, area.rs:14-81 */
typedef struct rt_light {
    uint32_t kind;
    uint8_t  color[3];
    uint8_t  shadows;             /* SUN only */
    double   intensity;
    int32_t  specular_power;
    uint32_t _pad;
    double   specular_strength;
    double   v[3];                /* SUN: vector; POINT: position; AREA sphere: centre */
    double   shadow_coefficient;  /* SUN only */
    double   max_distance;        /* POINT / AREA */
    /* AREA only */
    uint32_t area_surface;        /* RT_AREA_* */
    uint32_t area_iterations;     /* sample points per hit (area.rs:74-75, :117) */
    double   area_radius;         /* RT_AREA_SPHERE */
    double   area_corners[4][3];  /* RT_AREA_RECTANGLE: AreaSurface::Rectangle's array as it stands (area.rs:19), the
                                     order blerp takes its corners in (area.rs:41-48, math/mod.rs:22-26) */
} rt_light;

/* skybox.rs:15,25,35-44.  CUBEMAP: 4x3 atlas, cell = w/4 = h/3 (:49-51). */
typedef struct rt_skybox {
    uint32_t       kind;
    uint8_t        color[3];
    uint8_t        _pad;
    uint32_t       w, h;
    const uint8_t* rgb;
} rt_skybox;

typedef struct rt_scene_desc {
    rt_camera  camera;
    rt_options options;
    const rt_object*   objects;   uint32_t n_objects;   uint32_t _p0;
    const rt_sphere*   spheres;   uint32_t n_spheres;   uint32_t _p1;
    const rt_plane*    planes;    uint32_t n_planes;    uint32_t _p2;
    const rt_box*      boxes;     uint32_t n_boxes;     uint32_t _p3;
    const rt_mesh*     meshes;    uint32_t n_meshes;    uint32_t _p4;
    const rt_material* materials; uint32_t n_materials; uint32_t _p5;
    const rt_texture*  textures;  uint32_t n_textures;  uint32_t _p6;
    const rt_light*    lights;    uint32_t n_lights;    uint32_t _p7;
    rt_skybox  skybox;
} rt_scene_desc;

/* Ray classes.  One ray = one Scene::cast_ray_once call (scene.rs:79):
 * primary (scene.rs:216), reflection (:169,:190), transmission (:134,:154)
 * and shadow (sun.rs:89, point.rs:91).  The two same-object helper
 * intersects of the refraction branch (scene.rs:143-147) are not rays. */
typedef struct rt_stats {
    uint64_t rays_primary, rays_reflect, rays_transmit, rays_shadow;
    uint64_t rays_per_level[8];   /* trace_ray invocations per depth        */
    double   ms_upload;           /* last rt_scene_upload                   */
    double   ms_render;           /* first to last kernel of the last frame (HIP events on the main stream); a frame
                                   * rendered in several passes (large frames, see rt_render): the passes' sum, like the
                                   * other times and the counts below */
    double   ms_total;            /* render + copy-out wall time            */
    /* Per phase, the launches' own durations summed over depths (HIP events on the stream each phase runs on):
     * closest-hit casts; shadow casts; spawn + lighting; the final fold.  The phases of a frame overlap (the shadow
     * casts of a level run beside the trace of the next), so these add up to more than ms_render. */
    double   ms_trace, ms_shadow, ms_shade, ms_resolve;
    uint64_t device_bytes;        /* device memory held by the context      */
    /* Work the traversal kernels really did: inner BVH nodes visited (two box
     * tests each) and triangles tested.  Only counted when the context was
     * created with RTGPU_COUNT_VISITS=1 in the environment (instrumented kernel
     * variants, slower; "trace" / "shadow" / "shade" instrument one phase only);
     * 0 otherwise. */
    uint64_t nodes_visited, tris_tested;
    /* the same for the wave-coherent packet walks, counted once per wave: nodes a wave visited (it visits a node
     * when any of its 64 rays needs it); triangle steps = triangles a closest-hit packet tested, plus the rounds
     * of up to 64 (ray, leaf) pairs a shadow packet tested */
    uint64_t wave_nodes, wave_tris;
    uint32_t launches_trace, launches_shadow, launches_shade, _pad;   /* kernel launches of the last frame */
    /* rays whose walk exceeded its budget and was finished piecewise by the long-ray kernels */
    uint64_t long_trace, long_shadow;
    /* instrumented contexts only: the longest walk (inner nodes + triangle tests) of one lane / of one wave packet */
    uint64_t max_lane_steps, max_wave_steps;
    /* instrumented contexts only: bytes of BVH data (node and triangle records, at their real sizes) the walks of
     * the last frame fetched; a packet's shared fetch counts once */
    uint64_t bytes_fetched;
    /* The budget-free shadow packet kernel (the large launches of a frame, where most of its time goes): its own launch
     * durations summed over the levels that used it, their number, and (instrumented contexts) the BVH bytes those
     * levels' walks fetched. */
    double   ms_shadow_main;
    uint64_t bytes_fetched_main;
    uint32_t launches_shadow_main;
    /* 1: the frame was queued without reading the level sizes back (they were taken from the context's previous frame) */
    uint32_t speculative;
    /* frames of this context that were rendered twice because a level outgrew the room its predecessor suggested */
    uint64_t respeculated;
} rt_stats;

/* result of one probe ray (rt_cast_rays): the fields of object/mod.rs:18-36
 * plus the ids the reference's Hit lacks */
typedef struct rt_ray_hit {
    uint32_t object;              /* scene-order index, RT_ID_NONE on miss */
    uint32_t prim;                /* triangle index in the mesh's own order, RT_ID_NONE otherwise */
    double   near_t, far_t;
    double   normal[3], vnear[3], vfar[3];
    float    uv[2];
} rt_ray_hit;

typedef struct rt_ctx rt_ctx;

/* Creates a context on HIP device `device_id`.  Fails with RT_ERR_DEVICE when
 * no such device exists: there is no CPU fallback behind this library. */
int rt_ctx_create(int device_id, rt_ctx** out);
void rt_ctx_destroy(rt_ctx* ctx);

/* One context over n_devices GPUs of this node, for the reference's callers as they are: one process that asks for
 * one frame at a time (sdl/src/main.rs:61, :91 -> scene.rs:249-250 -> Scene::render, scene.rs:220).  The context is
 * used like any other -- rt_scene_upload puts the scene on every device, rt_render and rt_render_frames return whole
 * frames in host memory, rt_set_area_seed and rt_ctx_destroy reach every device -- and inside it device k renders
 * image-space shard k of n_devices (pixels are independent, scene.rs:239-242), the compact shard buffers are gathered
 * to device_ids[0] with RCCL (ncclGather, one communicator per device inside this process; librccl is opened when
 * the first such context is created) and device_ids[0] assembles the frame.  n_devices == 1 is rt_ctx_create: no
 * collective, no RCCL.  The entry points that take device pointers (rt_render_shard, rt_render_shard_batch,
 * rt_detile, rt_detile_frames) and the probes (rt_cast_rays, rt_cast_rays_noalloc) address device_ids[0] alone.
 * A list that names one GPU more than once (several shards on one GPU: a rehearsal on a small machine) is gathered
 * with device-to-device copies, RCCL having no such communicator. */
int rt_ctx_create_multi(const int* device_ids, int n_devices, rt_ctx** out);
int rt_ctx_device_count(const rt_ctx* ctx);   /* 1 for a context of rt_ctx_create, 0 for NULL */

/* Text of the last failure on this context (or of the last failed
 * rt_ctx_create when ctx is NULL).  Never NULL. */
const char* rt_last_error(const rt_ctx* ctx);

/* Deep-copies the scene to the device, re-emitting each mesh's top-down BVH
 * (acceleration.rs:407-448) in the device layout.  May be called again per
 * frame (sdl/src/main.rs:84-92 sequence mode).  A mesh tree too deep for the
 * traversal stacks (about 64 levels) is replaced by a balanced tree over the
 * same leaves in the same order: same hits, same winners between equal
 * distances.  Lights and area-light samples together (area_iterations per
 * area light, area.rs:117) may number 65535 at most: RT_ERR_UNSUPPORTED. */
int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* scene);

/* Seed of the area lights' sample points.  No counterpart in the reference:
 * lighting/area.rs:121 draws every sample point from an unseeded thread
 * generator, so two of its renders of one scene differ; here a sample point
 * is a function of (seed, pixel, the ray's place in the pixel's tree of rays,
 * light, sample number), the same on the device and in the CPU oracle, and a
 * frame is repeatable.  Applies to the scene in place and to later uploads.
 * Default 0x9E3779B97F4A7C15. */
int rt_set_area_seed(rt_ctx* ctx, uint64_t seed);

/* Replaces Scene::render (scene.rs:220-246).
 *   rgb_out     vw*vh*3 bytes, row-major, top-left origin (host memory)
 *   prim_id_out nullable; 2 u32 per pixel = (object index in scene order,
 *               triangle index in the mesh's own order) of the primary hit;
 *               RT_ID_NONE = miss / not a mesh
 *   stats_out   nullable
 * The device memory a frame's rays take goes with their number (about 290
 * bytes a ray).  A frame, or set of frames, that would take more than 32 GB is
 * rendered in several passes over interleaved parts of the image through the
 * same buffers, each pass writing its pixels in place: same frame, 1 % more
 * time where it applies (RTGPU_PASSES in the environment fixes the number). */
int rt_render(rt_ctx* ctx, uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out);

/* Image-space sharding for one-process-per-GPU runs (pixels are independent,
 * scene.rs:239-242).  The image is cut into supertiles of S x S tiles of
 * RT_TILE_W x RT_TILE_H pixels, S = rt_supertile() (4 unless the environment
 * variable RTGPU_SUPER_SHIFT = log2 S says otherwise; one value per process).
 * Supertiles are numbered row by row, each row rotated by a few columns
 * against the one above (so that the columns a shard owns differ from row to
 * row); shard s of n owns supertiles s, s+n, s+2n, ... of that numbering.
 * Which pixel a slot holds is the library's business: rt_detile puts gathered
 * shard buffers back into a row-major frame.  rt_shard_pixels returns the
 * number of pixel slots (whole supertiles, edges padded) shard s produces. */
#define RT_TILE_W 16
#define RT_TILE_H 16
int64_t rt_shard_pixels(int32_t vw, int32_t vh, uint32_t shard, uint32_t n_shards);
uint32_t rt_supertile(void);

/* Renders this shard's supertiles into DEVICE memory, compact: the shard's
 * k-th supertile is slots [k*S*S*256, (k+1)*S*S*256), tile by tile (row-major
 * inside the supertile), 256 slots per tile: four 8x8 pixel blocks (left to
 * right, top to bottom), each row-major.
 *   d_rgb   device pointer, rt_shard_pixels()*3 bytes
 *   d_ids   device pointer or NULL, rt_shard_pixels()*2 u32
 * The call returns after the kernels have completed. */
int rt_render_shard(rt_ctx* ctx, uint32_t shard, uint32_t n_shards,
                    void* d_rgb, void* d_ids, rt_stats* stats_out);

/* Scatters gathered compact shard buffers (shard-major, each padded to
 * slots_per_shard slots of `elem_bytes` bytes) into a row-major image, both in
 * DEVICE memory.  elem_bytes is 3 (rgb) or 8 (id pairs). */
int rt_detile(rt_ctx* ctx, const void* d_gathered, uint32_t n_shards,
              int64_t slots_per_shard, uint32_t elem_bytes, void* d_image);

/* The sequence workload (sdl/src/main.rs:70-95 renders frame after frame) on a GPU that owns one shard of every
 * frame: renders shard `shard` of n_frames frames that differ in their camera only -- a camera path through the
 * uploaded scene -- as ONE set of launches, so that the launches of a small shard stay large enough to fill the
 * chip.  cameras[f].vw/vh must equal the uploaded scene's; cameras == NULL: every frame through the uploaded scene's
 * own camera (a still held for n_frames frames).  Frame f's compact shard (layout as rt_render_shard)
 * starts at slot f * frame_stride of d_rgb / d_ids; frame_stride >= rt_shard_pixels() (ranks pad to the largest
 * shard, shard 0's, so that gathered buffers line up: rt_detile_frames).  1 <= n_frames <= RT_BATCH_MAX. */
#define RT_BATCH_MAX 64u
int rt_render_shard_batch(rt_ctx* ctx, uint32_t shard, uint32_t n_shards, uint32_t n_frames, const rt_camera* cameras,
                          int64_t frame_stride, void* d_rgb, void* d_ids, rt_stats* stats_out);

/* rt_detile for gathered batches: d_gathered is [shard][frame][slot] (each run padded to slots_per_shard slots),
 * d_images receives n_frames row-major images one after the other. */
int rt_detile_frames(rt_ctx* ctx, const void* d_gathered, uint32_t n_shards, uint32_t n_frames,
                     int64_t slots_per_shard, uint32_t elem_bytes, void* d_images);

/* The sequence workload behind the reference's own entry point (sdl/src/main.rs:84-92: frame after frame, each
 * through Scene::render, scene.rs:220): n_frames frames of a camera path through the uploaded scene, rendered as one
 * set of launches on every device of the context and returned as whole frames in HOST memory, one after the other
 * (rgb_out: n_frames * vw*vh*3 bytes; prim_id_out: nullable, n_frames * vw*vh * 2 u32).  cameras[f].vw/vh must equal
 * the uploaded scene's; cameras == NULL holds the scene's own view.  1 <= n_frames <= RT_BATCH_MAX.  On a context
 * over several GPUs each device renders its shard of every frame and one gather moves the whole set. */
int rt_render_frames(rt_ctx* ctx, uint32_t n_frames, const rt_camera* cameras, uint8_t* rgb_out, uint32_t* prim_id_out,
                     rt_stats* stats_out);

/* How a context over n_devices GPUs lays out a set of n_frames frames (no GPU needed; the library uses these figures
 * itself): every device contributes the same count to the gather, its shard padded to the largest one (shard 0's).
 *   slots_per_shard     rt_shard_pixels(vw, vh, 0, n_devices)
 *   part_bytes_*        one device's compact buffer: [frame][slot]
 *   gathered_bytes_*    on the first device: [shard][frame][slot], device k's part at k * part_bytes
 *   frame_bytes_rgb     one assembled frame */
typedef struct rt_multi_plan {
    int64_t slots_per_shard;
    int64_t part_bytes_rgb, part_bytes_ids;
    int64_t gathered_bytes_rgb, gathered_bytes_ids;
    int64_t frame_bytes_rgb;
} rt_multi_plan;
int rt_multi_layout(int32_t vw, int32_t vh, uint32_t n_devices, uint32_t n_frames, rt_multi_plan* out);

/* Probe: casts n arbitrary rays through the uploaded scene with the semantics
 * of Scene::cast_ray_once (scene.rs:79-82); host pointers.
 * origins/dirs: n*3 doubles each. */
int rt_cast_rays(rt_ctx* ctx, uint32_t n, const double* origins, const double* dirs,
                 rt_ray_hit* hits_out);

/* rt_cast_rays for callers that must not allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  (The call copies from and to pageable host memory and waits for
 * the context's stream, so it cannot itself be recorded into a HIP graph.)  Same arguments and results as
 * rt_cast_rays, but the rays and hit records are staged in a fixed array that lives in the library's device image
 * (RT_CAST_NOALLOC_MAX * 152 bytes, about 10 MB, held on every GPU the library runs on until the process ends), so
 * the call makes no hipMalloc / hipFree.  n above RT_CAST_NOALLOC_MAX is refused with RT_ERR_INVALID.  All contexts
 * on one GPU share that array: calls that overlap in time on one GPU race with each other. */
#define RT_CAST_NOALLOC_MAX 65536u
int rt_cast_rays_noalloc(rt_ctx* ctx, uint32_t n, const double* origins, const double* dirs,
                         rt_ray_hit* hits_out);

#ifdef __cplusplus
}
#endif
#endif /* RTGPU_H */
