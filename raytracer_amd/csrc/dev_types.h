// Device-side data layout of an uploaded scene (gfx950).  All geometry stays
// f64 -- the reference path is f64 end to end (raytracer/src/math/vector.rs:20-24)
// and its 1e-11 ray offsets (scene.rs:13) vanish in f32.
#pragma once
#include <stdint.h>

#define RT_EPSILON 0.00000000001   /* scene.rs:13 */
#define RT_METER 2.                /* lighting/mod.rs:13 */
#define RT_NONE 0xFFFFFFFFu
#define RT_MAX_STACK 64            /* deepest mesh BVH the traversal stack holds */
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 8             /* stack entries per lane kept in LDS (16 KB per workgroup, so that LDS does not cap the waves per SIMD); deeper ones spill to private memory */
#endif

struct D3 { double x, y, z; };

// One inner node of the device BVH: the reference's binary top-down tree (acceleration.rs:407-417) re-emitted
// four wide -- each node adopts its grandchildren -- so that a ray makes about half as many dependent node
// fetches.  One 128-byte record carries the boxes and references of all (up to) four children, plane by
// plane (min x of the 4 children, min y, ...), so a visit is seven 16-byte loads and tests the four boxes
// without touching the children themselves.  The boxes are f32, rounded OUTWARD from the reference's f64
// slab bounds (centroid -+ size, acceleration.rs:76,135): they are used for culling only -- a box test that
// passes in the reference passes here -- while every triangle test stays f64 and exact.  Leaves keep the
// reference's contents and order.
#define RT_LEAF_BIT 0x80000000u
#define RT_NO_CHILD 0xFFFFFFFFu
struct __attribute__((aligned(16))) DNode {
    float minx[4], miny[4], minz[4];
    float maxx[4], maxy[4], maxz[4];
    uint32_t ref[4];          // inner child: node index; leaf child: RT_LEAF_BIT | first triangle (leaf order); RT_NO_CHILD: empty slot
    uint32_t pad[4];
};

// One triangle in leaf order (the order mesh.rs:326-345 gathers candidates):
// v0 and the two edges triangle_intersect forms first (mesh.rs:26-27), the
// triangle's index in the mesh's own order, and whether it ends its leaf.
struct __attribute__((aligned(16))) DTri { double v0[3], e1[3], e2[3]; uint32_t orig, last; };

struct DMesh {
    const DNode* nodes;        // inner nodes; nodes[0] is the root when n_inner > 0
    const DTri* tris;          // leaf order
    const double* normals;     // [n][3]
    const float* texcoords;    // [n][2]
    const uint32_t* tri_nrm;   // [n_tris][3], mesh order
    const uint32_t* tri_uv;    // [n_tris][3] or null
    float root_min[3], root_max[3];   // the root's own box, f32 rounded outward
    float coord_max;                  // largest |coordinate| of any box of the mesh (bounds the f32 slab rounding error)
    uint32_t _pad0;
    uint32_t n_inner, n_tris, n_texcoords;
    uint32_t root_ref;         // 0 (inner node 0), or RT_LEAF_BIT | 0 when the root itself is a leaf
};

struct DObject {
    uint32_t kind, material, payload;
    uint32_t need_far;         // mesh: 1 = the material needs Hit::far (transparency with ior == 1);
                               // sphere / box: RT_IN_TLAS = reached through the object tree, skipped by the linear scans
};
#define RT_IN_TLAS 2u

// Object tree over the bounded analytic objects (spheres, boxes) of a scene with many of them: the reference tests
// every object per cast (scene.rs:57-82), which is 1728 sphere tests per ray in scenes/huge_grid.sdl.  Binary tree in
// depth-first order with skip links, so a walk needs no stack: a node that is missed continues at `skip`, a node that
// is entered at the next record.  Leaves hold one object.  Boxes are f32, rounded outward and inflated.
struct DTlasNode {
    float bmin[3], bmax[3];
    uint32_t skip;             // next record when this subtree is skipped or done
    uint32_t object;           // leaf: index into DScene::objects; inner node: RT_TLAS_INNER | index of its second child
                               // (the first child is the next record)
};
#define RT_TLAS_INNER 0x80000000u

// The same tree for the walks that own a stack (the lane-by-lane trace kernels), folded four-wide: one 128-byte record
// holds the boxes of up to four children (an inner node's two children, each replaced by its own two where it has
// them), so a step of the walk covers two levels of the binary tree with one dependent fetch, and leaves need no
// record at all.  A child reference is RT_TLAS_LEAF | object index, the index of the child's own record, or
// RT_TLAS_NONE for an unused slot.  For a leaf child `what` says what DScene::objects would (RT_TLAS_BOX | index into
// DScene::boxes, or the index into DScene::spheres), which saves the dependent fetch of the object record.
struct __attribute__((aligned(16))) DTlasQuad {
    float minx[4], miny[4], minz[4];
    float maxx[4], maxy[4], maxz[4];
    uint32_t ref[4];
    uint32_t what[4];
};
#define RT_TLAS_LEAF 0x80000000u
#define RT_TLAS_NONE 0xFFFFFFFFu
#define RT_TLAS_BOX 0x40000000u

// The four-wide records once more, in 64 bytes, for the walks in which every lane fetches records of its own (the
// lane-by-lane closest-hit and shadow casts): those walks are bound by the number of cache-line look-ups their loads
// make -- 64 different lines per load instruction -- and not by arithmetic, so the boxes are kept as half-precision
// floats, rounded outward (still conservative: a box only ever grows), and a record is four 16-byte loads instead
// of eight.  A leaf child is RT_TLAS_LEAF | (RT_TLAS_BOX for a cube) | object index, and the object's own numbers
// stand in DScene::tlas_geom at 6 doubles per object index (sphere: centre, radius; cube: position, half extent), so
// that neither the object record nor a `what` word is fetched.
struct __attribute__((aligned(16))) DTlasHalf {
    uint16_t minx[4], miny[4], minz[4];
    uint16_t maxx[4], maxy[4], maxz[4];
    uint32_t ref[4];
};

struct DMaterial {
    uint32_t tex_kind;
    uint32_t tex_w, tex_h;
    uint32_t a_rgb, b_rgb;     // packed r | g<<8 | b<<16
    uint32_t _pad;
    const uint8_t* tex_rgb;
    double reflectiveness, transparency, ior, emissivity;
};

struct DLight {
    uint32_t kind;
    int32_t specular_power;
    uint32_t shadows, _pad;
    double col[3];             // colour / 255 (vector.rs:216-224)
    double intensity, specular_strength;
    double v[3];               // sun: vector; point: position; area sphere: centre
    double shadow_coefficient, max_distance;
    // An area light (area.rs:56-81) takes one record per sample point, in a row: the record of sample `area_it` of
    // `area_n`, all with the light's own index `area_li` in the scene's list (kind RT_DLIGHT_AREA_SAMPLE), so that the
    // shadow kernels find one cast per record as for the other lights.  A light of no samples takes one record of kind
    // RT_DLIGHT_AREA_EMPTY.
    uint32_t area_surface, area_it, area_n, area_li;
    double area_radius;
    double area_corners[4][3];
};
enum { RT_DLIGHT_SUN = 0, RT_DLIGHT_POINT = 1, RT_DLIGHT_AREA_SAMPLE = 2, RT_DLIGHT_AREA_EMPTY = 3 };

// camera (camera.rs:55-67): rot = from_euler_xyz(-pitch, yaw, 0) hoisted to the host
struct DCamera {
    double cam_o[3];
    double rot[3][3];
    double half_w, half_h, cam_z;   // vw*0.5, vh*0.5, vh*0.5*chf
};

struct DScene {
    DCamera cam;
    int32_t vw, vh;
    uint32_t max_depth;
    uint32_t n_objects, n_lights, n_meshes;
    double ambient[3];              // ambient / 255
    const DObject* objects;
    const double* spheres;          // [n][4] c, r
    const double* planes;           // [n][8] p0, n, uv_wrap (as double holding the f32 value), pad
    const double* boxes;            // [n][6] pos, half
    const DMesh* meshes;
    const DTlasNode* tlas;          // null when the scene has few analytic objects
    const DTlasQuad* tlas_quad;     // the same tree folded four-wide (record 0 = the root)
    const DTlasHalf* tlas_half;     // ... and in 64-byte records with half-precision boxes, same numbering
    const double* tlas_geom;        // [n_objects][6]: the tree's spheres and cubes by object index
    float tlas_half_coord_max;      // the largest |coordinate| of any half-precision box
    uint32_t _pad_tlas;
    DMesh tlas_mesh;                // the four-wide tree described like a mesh's (nodes, root box, coord_max, root_ref = 0): what the packet walks take
    const uint32_t* linear;         // indices of the objects the casts scan one by one (all that are not in the tree)
    uint32_t n_tlas, n_linear;
    const DMaterial* materials;
    const DLight* lights;
    uint32_t sky_kind, sky_rgb;     // packed solid colour
    uint32_t sky_w, sky_h;
    const uint8_t* sky_pix;
    // area lights: whether the scene has any (only then do rays carry keys), and the seed of their sample points
    unsigned long long area_seed;
    uint32_t has_area, _pad_area;
};

// Structure-of-arrays state of one wavefront level: every ray of one
// trace_ray depth (scene.rs:85) across the frame.
struct DLevel {
    double* ro[3]; double* rd[3];                 // the ray
    // closest hit, as k_trace leaves it: the Hit of object/mod.rs:18-36 minus the two points (vnear = o + d * near and
    // vfar = o + d * far are recomputed where needed: one multiply-add per component, the reference's own expression)
    uint32_t* hit_obj; uint32_t* hit_prim;        // object; triangle in the mesh's own order (RT_NONE for other objects)
    double* hit_t; double* hit_t2;                // Hit::near, Hit::far
    float* hit_u; float* hit_v;                   // Hit::uv
    double* nrm[3];                               // Hit::normal
    double* lit[3];                               // lit colour before the child mixes (scene.rs:122); shares nrm's memory
    double* w_t; double* w_r; double* emis;       // mix weights
    uint32_t* child0; uint32_t* child1;           // index in the next level (transmission, reflection)
    uint32_t* base_flags;                         // base colour rgb | flags << 24
    uint32_t* result;                             // final colour of this trace_ray call, packed rgb
    unsigned long long* key;                      // what the ray's area-light sample points are drawn from (written in scenes with area lights only)
};

enum { NF_DONE = 1, NF_TRANSPARENT = 2, NF_T_REFLECT = 4, NF_REFLECT = 8, NF_EMISSIVE = 16 };

// Frame counters.  Every level has its own pair of allocation cursors (one atomic per workgroup each), so that the
// kernels of different levels can be in flight together; the statistics are spread over
// RT_SHARDS cache lines (a workgroup adds to shard blockIdx % RT_SHARDS) because thousands of adds to ONE word
// serialise at about 12 ns each and would cost milliseconds per frame.  The host sums the shards.
#define RT_SHARDS 64
struct DShard { unsigned long long rays_reflect, rays_transmit, rays_shadow, nodes_visited, tris_tested, wave_nodes, wave_tris, max_lane_steps, max_wave_steps, units_fetched, pad[6]; };   // 128 B
struct DLevelCounts {
    unsigned long long next_count;   // rays this level appended to the next one
    unsigned long long hit_count;    // lit hits of this level (length of its k_shadow work list)
    unsigned long long long_trace;   // rays whose closest-hit walk gave up (entries of the level's long-ray list)
    unsigned long long long_shadow;  // shadow rays whose wave gave up
    unsigned long long child_bound;  // upper bound of next_count, known before the level spawns (k_count_children)
    unsigned long long pad[11];
};
struct DCounters { DShard shard[RT_SHARDS]; };
