/*
 * oracle.cpp -- CPU restatement of the reference's per-pixel render loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against; it is never linked into, imported by or called from the product
 * libraries (librtgpu.so / librthost.so).  Only tests/, the smoke check and
 * the cpu_baseline leg of bench.py load it.
 *
 * HOW IT IS PINNED: the reference (a Rust workspace needing a nightly
 * toolchain) cannot be built in this environment and ships no tests.  What it
 * does ship is its author's own renders, with their scene listings, in the
 * README: this restatement reproduces five of them to the byte -- the
 * 16-sphere ring (images/readme/10_19_2021.png) and the icospheres of
 * scenes/icosahedron.sdl at 0, 2 and 5 subdivisions (11_22_2021_1..3.png:
 * mesh BVH, triangle test, normals), and all but 14 of 2,073,600 pixels of
 * the cube / mirror sphere / checkered plane / point light scene
 * (10_7_2021.png, whose cube position is random() and was fitted); and a
 * sixth in distribution, the Cornell box of scenes/cornell.sdl
 * (11_22_2021_5.png: glass sphere, emissive panel, an area light whose sample
 * points the reference draws from an unseeded generator, so that no two
 * renders are equal): the image is indistinguishable from a further render
 * of this file with another seed, value by value, on the glass sphere alone
 * as well, and no longer so with the sphere's index of refraction at 1.45;
 * likewise the two Cornell boxes of scenes/cornell_icosahedron.sdl
 * (11_22_2021_6.png, _7.png: a mesh under the area light) --
 * and the sky of two more (images/readme/10_4_2021.png, 10_4_2021_2.png:
 * 1,574,249 pixels seen through the cubemap of assets/storforsen_4.jpg) to
 * the byte, which pins Cubemap::ray_color / poll_tex for the -z and +-x
 * cells, the camera's rotation and -- through the blend of four texels per
 * pixel -- the JPEG decoder's bytes (tests/test_cubemap_golden.py) --
 * plus hand-derived analytic known-answer tests (tests/test_oracle_kat.py).
 * PARITY UNPINNED for what none of those images shows: the +-y cells of the
 * cubemap and their flips (skybox.rs:102-116), texture coordinates on meshes
 * (mesh.rs:61-77), normals averaged over shared vertices (mesh.rs:208-254),
 * refraction through meshes and cubes; these rest on the known-answer tests
 * and the cited lines.
 *
 * Every function cites the reference lines it follows.  Rules kept throughout:
 * f64 where the reference is f64, f32 exactly where it is f32, no fused
 * multiply-add (build with -ffp-contract=off), Rust's f64::max/min = fmax/fmin,
 * signum(+-0) = +-1, saturating float->int casts with NaN -> 0, stable sorts.
 */
#include "../include/rtgpu.h"
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr double EPSILON = 0.00000000001;  // scene.rs:13
constexpr double METER = 2.;               // lighting/mod.rs:13

/* ---- per-thread work counters (oracle extension, not in the reference) -- */
enum { CTX_CLOSEST = 0, CTX_SHADOW = 1, CTX_HELPER = 2 };   // which kind of cast a primitive test belongs to
struct Counts {
    uint64_t node_tests = 0, tri_tests = 0, sphere_tests = 0, plane_tests = 0, box_tests = 0;   // all contexts
    uint64_t by_ctx[3][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};             // [ctx][node,tri,sphere,plane,box]
    int ctx = CTX_CLOSEST;
    uint64_t texel_image = 0, texel_cubemap = 0;
    uint64_t rays_primary = 0, rays_reflect = 0, rays_transmit = 0, rays_shadow = 0;
    uint64_t rays_per_level[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t would_panic = 0;
    void add(const Counts& o) {
        node_tests += o.node_tests; tri_tests += o.tri_tests; sphere_tests += o.sphere_tests;
        plane_tests += o.plane_tests; box_tests += o.box_tests;
        for (int c = 0; c < 3; c++) for (int k = 0; k < 5; k++) by_ctx[c][k] += o.by_ctx[c][k];
        texel_image += o.texel_image; texel_cubemap += o.texel_cubemap;
        rays_primary += o.rays_primary; rays_reflect += o.rays_reflect;
        rays_transmit += o.rays_transmit; rays_shadow += o.rays_shadow;
        for (int i = 0; i < 8; i++) rays_per_level[i] += o.rays_per_level[i];
        would_panic += o.would_panic;
    }
};
thread_local Counts* tl_counts = nullptr;
#define COUNT(field) do { if (tl_counts) tl_counts->field++; } while (0)
#define COUNT_TEST(field, k) do { if (tl_counts) { tl_counts->field++; tl_counts->by_ctx[tl_counts->ctx][k]++; } } while (0)
struct CtxScope {   // switches the counting context for the duration of a cast
    int saved = 0;
    explicit CtxScope(int c) { if (tl_counts) { saved = tl_counts->ctx; tl_counts->ctx = c; } }
    ~CtxScope() { if (tl_counts) tl_counts->ctx = saved; }
};

/* ---- math/vector.rs ------------------------------------------------------ */
struct V3 { double x, y, z; };
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }      // vector.rs:116-126
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }      // :136-146
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }      // :148-158
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }        // :160-170
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }      // :180-190
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }        // :192-202
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }                           // :204-214
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // :60-62
inline V3 cross(V3 a, V3 b) {                                                      // :65-71
    return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
inline double magnitude(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // :74-76 (powi(2) == x*x)
inline V3 normalize(V3 a) { return a / magnitude(a); }                             // :79-81
inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }  // :93-99

/* Rust f64::signum: NaN -> NaN, else copysign(1, x) (so signum(+-0) = +-1). */
inline double signum(double x) { return std::isnan(x) ? x : std::copysign(1.0, x); }
/* Rust f64::clamp(lo, hi): NaN stays NaN. */
inline double clampd(double x, double lo, double hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
inline float clampf(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
/* Rust `as u8` / `as u32` from float: saturating, NaN -> 0, truncation toward zero. */
inline uint8_t as_u8(double v) { if (!(v > 0.0)) return 0; if (v >= 255.0) return 255; return (uint8_t)v; }
inline uint32_t as_u32(double v) { if (!(v > 0.0)) return 0; if (v >= 4294967295.0) return 4294967295u; return (uint32_t)v; }
inline uint32_t as_u32f(float v) { if (!(v > 0.0f)) return 0; if (v >= 4294967296.0f) return 4294967295u; return (uint32_t)v; }
/* Rust f32::rem_euclid */
inline float rem_euclid_f(float a, float b) { float r = std::fmod(a, b); return r < 0.0f ? r + std::fabs(b) : r; }
/* Rust f64::powi lowers to compiler-rt __powidf2 for a run-time exponent. */
inline double powi(double a, int b) {
    const bool recip = b < 0;
    double r = 1;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}
inline double lerp(double a, double b, double c) { return a + (b - a) * c; }       // math/mod.rs:17-19
inline V3 vlerp(V3 a, V3 b, double t) { return {lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)}; }  // vector.rs:226-234
inline V3 blerp(double tx, double ty, V3 c00, V3 c10, V3 c01, V3 c11) {            // math/mod.rs:22-26
    V3 a = c00 * (1. - tx) + c10 * tx;
    V3 b = c01 * (1. - tx) + c11 * tx;
    return a * (1. - ty) + b * ty;
}

/* ---- material.rs Color ----------------------------------------------------- */
struct Color { uint8_t r, g, b; };
inline Color newf(double r, double g, double b) {                                  // material.rs:38-44
    return {as_u8(clampd(r, 0., 1.) * 255.0), as_u8(clampd(g, 0., 1.) * 255.0), as_u8(clampd(b, 0., 1.) * 255.0)};
}
inline Color to_color(V3 v) { return newf(v.x, v.y, v.z); }                        // :78-82
inline V3 to_v3(Color c) { return {c.r / 255., c.g / 255., c.b / 255.}; }          // vector.rs:216-224
inline Color from_normal(V3 n) { return newf(n.x / 2. + 0.5, n.y / 2. + 0.5, n.z / 2. + 0.5); }  // material.rs:47-49

/* ---- math/ray.rs ------------------------------------------------------------ */
struct Ray { V3 origin, direction, m; };
inline Ray ray_new(V3 o, V3 d) { return {o, d, {1. / d.x, 1. / d.y, 1. / d.z}}; }  // ray.rs:17-23, vector.rs:84-90
inline V3 along(const Ray& r, double t) { return r.origin + r.direction * t; }     // ray.rs:26-28
inline Ray reflect(const Ray& r, V3 pos, V3 n) {                                   // ray.rs:43-46
    V3 dir = r.direction - n * (2. * dot(r.direction, n));
    return ray_new(pos, dir);
}

/* math/mod.rs:48-63 */
inline bool refraction_vec(const Ray& in_ray, V3 normal, double from_ior, double to_ior, V3& out) {
    double n = from_ior / to_ior;
    double cos_i = -dot(normal, in_ray.direction);
    double sin_t2 = n * n * (1. - cos_i * cos_i);
    if (sin_t2 > 1.) return false;
    double cos_t = std::sqrt(1. - sin_t2);
    out = in_ray.direction * n + normal * (n * cos_i - cos_t);
    return true;
}

/* ---- math/matrix.rs ----------------------------------------------------------- */
struct Matrix { double x, y, z, m00, m01, m02, m10, m11, m12, m20, m21, m22; };
inline Matrix mat_mul(const Matrix& A, const Matrix& B) {                          // matrix.rs:148-185
    double a[4][4] = {{A.m00, A.m01, A.m02, A.x}, {A.m10, A.m11, A.m12, A.y}, {A.m20, A.m21, A.m22, A.z}, {0., 0., 0., 1.}};
    double b[4][4] = {{B.m00, B.m01, B.m02, B.x}, {B.m10, B.m11, B.m12, B.y}, {B.m20, B.m21, B.m22, B.z}, {0., 0., 0., 1.}};
    double o[4][4] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) o[i][j] += a[i][k] * b[k][j];
    return {o[0][3], o[1][3], o[2][3], o[0][0], o[0][1], o[0][2], o[1][0], o[1][1], o[1][2], o[2][0], o[2][1], o[2][2]};
}
inline Matrix from_forward(V3 vec) {                                               // matrix.rs:38-57
    V3 forward = vec;
    V3 right = cross(v3(0., 1., 0.), vec);
    V3 up = cross(forward, right);
    return {0., 0., 0., right.x, right.y, right.z, up.x, up.y, up.z, -forward.x, -forward.y, -forward.z};
}
/* The sine and cosine of one angle come from ONE sincos() call: an optimising compiler turns a sin/cos pair of the same
 * argument into that call anyway (gcc here, LLVM for the reference's release build), glibc's sincos and sin can differ
 * in the last bit, and the device library's host code must arrive at the same matrix whatever its compiler does. */
inline Matrix from_euler_xyz(double x, double y, double z) {                       // matrix.rs:60-72
    double sx, cx, sy, cy, sz, cz;
    ::sincos(x, &sx, &cx); ::sincos(y, &sy, &cy); ::sincos(z, &sz, &cz);
    Matrix a = {0., 0., 0., 1., 0., 0., 0., cx, -sx, 0., sx, cx};
    Matrix b = {0., 0., 0., cy, 0., sy, 0., 1., 0., -sy, 0., cy};
    Matrix c = {0., 0., 0., cz, -sz, 0., sz, cz, 0., 0., 0., 1.};
    return mat_mul(mat_mul(a, b), c);
}

/* camera.rs:55-67 */
inline V3 direction_fov(const rt_camera& cam, double x, double y) {
    double nx = x - (double)cam.vw * 0.5;
    double ny = y - (double)cam.vh * 0.5;
    double z = (double)cam.vh * 0.5 * cam.chf;
    return normalize(v3(nx, -ny, -z));
}
inline V3 direction_at(const rt_camera& cam, double x, double y) {
    Matrix m = mat_mul(from_forward(direction_fov(cam, x, y)), from_euler_xyz(-cam.pitch, cam.yaw, 0.));
    return v3(-m.m20, -m.m21, -m.m22);                                             // matrix.rs:113-115
}

/* ---- object/mod.rs:18-36 (+ the ids the reference's Hit lacks) ------------------ */
struct Hit {
    V3 normal; double near; V3 vnear; double far; V3 vfar; float u, v;
    uint32_t prim = RT_ID_NONE;
};

/* sphere.rs:33-62 */
bool sphere_intersect(const rt_sphere& s, const Ray& ray, Hit& h) {
    COUNT_TEST(sphere_tests, 2);
    V3 origin = v3(s.c[0], s.c[1], s.c[2]);
    double rad2 = s.r * s.r;
    V3 l = origin - ray.origin;
    double t2 = dot(l, ray.direction);
    if (t2 < 0.0) return false;
    double d2 = dot(l, l) - t2 * t2;
    if (d2 > rad2) return false;
    double t3 = std::sqrt(rad2 - d2);
    double t0 = t2 - t3;
    double t1 = t2 + t3;
    V3 vtn = along(ray, t0);
    V3 vtf = along(ray, t1);
    V3 norm = normalize(vtn - origin);
    const float PI = 3.14159265358979323846f;                                      // std::f32::consts::PI
    h.normal = norm; h.near = t0; h.vnear = vtn; h.far = t1; h.vfar = vtf;
    h.u = 0.5f + (float)std::atan2(norm.x, norm.z) / (PI * 2.f);
    h.v = 0.5f - (float)std::asin(norm.y) / PI;
    h.prim = RT_ID_NONE;
    return true;
}

/* plane.rs:49-79 */
bool plane_intersect(const rt_plane& p, const Ray& ray, Hit& h) {
    COUNT_TEST(plane_tests, 3);
    V3 origin = v3(p.p0[0], p.p0[1], p.p0[2]);
    V3 normal = v3(p.n[0], p.n[1], p.n[2]);
    double denom = dot(normal, ray.direction);
    if (std::fabs(denom) > EPSILON) {
        double t = dot(origin - ray.origin, normal) / denom;
        if (t > 0.) {
            V3 pt = along(ray, t);
            h.normal = normal * -signum(denom);
            h.near = t; h.vnear = pt; h.far = t; h.vfar = pt;
            if (normal.y != 0.) {
                h.u = rem_euclid_f((float)pt.x / p.uv_wrap, 1.f);
                h.v = rem_euclid_f((float)pt.z / p.uv_wrap, 1.f);
            } else {
                h.u = 0.f; h.v = 0.f;
            }
            h.prim = RT_ID_NONE;
            return true;
        }
        return false;
    }
    return false;
}

/* aabb.rs:18-67 */
bool box_intersect(V3 pos, V3 size, const Ray& ray, Hit& h) {
    COUNT_TEST(box_tests, 4);
    V3 ro = ray.origin - pos;
    V3 s = v3(-signum(ray.direction.x), -signum(ray.direction.y), -signum(ray.direction.z));
    V3 t1 = ray.m * (-ro + (s * size));
    V3 t2 = ray.m * (-ro - (s * size));
    double tn = std::fmax(std::fmax(t1.x, t1.y), t1.z);
    double tf = std::fmin(std::fmin(t2.x, t2.y), t2.z);
    if (tn > tf || tf < 0.) return false;
    V3 normal;
    if (t1.x > t1.y && t1.x > t1.z) normal = v3(s.x, 0., 0.);
    else if (t1.y > t1.z) normal = v3(0., s.y, 0.);
    else normal = v3(0., 0., s.z);
    V3 pn = along(ray, tn);
    V3 pf = along(ray, tf);
    V3 pns = (v3(1., 1., 1.) - vabs(normal)) * (pn - pos) / size;
    double u0, u1;
    if (normal.y == 1.) { u0 = pns.x; u1 = pns.z; }
    else if (normal.y == -1.) { u0 = -pns.x; u1 = -pns.z; }
    else if (normal.x == 1.) { u0 = -pns.z; u1 = -pns.y; }
    else if (normal.x == -1.) { u0 = pns.z; u1 = -pns.y; }
    else if (normal.z == 1.) { u0 = pns.x; u1 = -pns.y; }
    else if (normal.z == -1.) { u0 = -pns.x; u1 = -pns.y; }
    else { u0 = 0.; u1 = 0.; }
    h.normal = normal; h.near = tn; h.vnear = pn; h.far = tf; h.vfar = pf;
    h.u = (float)u0 * 0.5f + 0.5f;
    h.v = (float)u1 * 0.5f + 0.5f;
    h.prim = RT_ID_NONE;
    return true;
}

/* acceleration.rs:134-151 (centroid = (min+max)*0.5, :76) */
bool aabb_bool(const rt_bvh_node& nd, const Ray& ray) {
    COUNT_TEST(node_tests, 0);
    V3 mn = v3(nd.min[0], nd.min[1], nd.min[2]);
    V3 mx = v3(nd.max[0], nd.max[1], nd.max[2]);
    V3 centroid = (mn + mx) * 0.5;
    V3 size = mx - centroid;
    V3 nro = centroid - ray.origin;
    V3 s = v3(-signum(ray.direction.x), -signum(ray.direction.y), -signum(ray.direction.z));
    V3 ssize = s * size;
    V3 t1 = ray.m * (nro + ssize);
    V3 t2 = ray.m * (nro - ssize);
    double tn = std::fmax(std::fmax(t1.x, t1.y), t1.z);
    double tf = std::fmin(std::fmin(t2.x, t2.y), t2.z);
    return !(tn > tf || tf < 0.);
}

/* mesh.rs:12-59 */
struct TriIntersect { V3 p; double t; float u, v; };
bool triangle_intersect(V3 v0, V3 v1, V3 v2, const Ray& ray, TriIntersect& out) {
    COUNT_TEST(tri_tests, 1);
    V3 edge1 = v1 - v0;
    V3 edge2 = v2 - v0;
    V3 h = cross(ray.direction, edge2);
    double a = dot(edge1, h);
    if (a > -EPSILON && a < EPSILON) return false;
    double f = 1. / a;
    V3 s = ray.origin - v0;
    double u = f * dot(s, h);
    if (!(u >= 0.0 && u <= 1.0)) return false;
    V3 q = cross(s, edge1);
    double v = f * dot(ray.direction, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(edge2, q);
    if (t > EPSILON) {
        out.p = along(ray, t); out.t = t; out.u = (float)u; out.v = (float)v;
        return true;
    }
    return false;
}

inline V3 mesh_vert(const rt_mesh& m, uint32_t i) { return v3(m.verts[3 * i], m.verts[3 * i + 1], m.verts[3 * i + 2]); }
inline V3 mesh_normal(const rt_mesh& m, uint32_t i) { return v3(m.normals[3 * i], m.normals[3 * i + 1], m.normals[3 * i + 2]); }

/* mesh.rs:326-345: depth-first, a then b; a node contributes only if its own
 * box is hit; every reached leaf contributes all its indices. */
void sbvh_intersection(const rt_mesh& m, uint32_t node, const Ray& ray, std::vector<uint32_t>& out) {
    const rt_bvh_node& nd = m.nodes[node];
    if (!aabb_bool(nd, ray)) return;
    if (nd.count == 0) {
        sbvh_intersection(m, node + 1, ray, out);
        sbvh_intersection(m, nd.right_or_first, ray, out);
    } else {
        for (uint32_t k = 0; k < nd.count; k++) out.push_back(m.prim_indices[nd.right_or_first + k]);
    }
}

/* mesh.rs:348-439 */
bool mesh_intersect(const rt_mesh& m, const Ray& ray, Hit& h) {
    if (m.n_nodes == 0) return false;   /* reference: assert!(sbvh.is_some()) :350 */
    std::vector<uint32_t> tris;
    sbvh_intersection(m, 0, ray, tris);
    if (tris.empty()) return false;
    struct Cand { uint32_t i; TriIntersect h; };
    std::vector<Cand> hits;
    for (uint32_t i : tris) {
        const uint32_t* t = &m.tri_pos[3 * i];
        TriIntersect ti;
        if (triangle_intersect(mesh_vert(m, t[0]), mesh_vert(m, t[1]), mesh_vert(m, t[2]), ray, ti)) hits.push_back({i, ti});
    }
    std::stable_sort(hits.begin(), hits.end(), [](const Cand& a, const Cand& b) { return a.h.t < b.h.t; });   // :374-376
    if (hits.empty()) return false;
    const Cand& t = hits[0];
    /* triangle_intersect_normal, mesh.rs:79-89: weights are f32, widened */
    {
        const uint32_t* n = &m.tri_nrm[3 * t.i];
        float u = 1.f - t.h.u - t.h.v, v = t.h.u, w = t.h.v;
        h.normal = mesh_normal(m, n[0]) * (double)u + mesh_normal(m, n[1]) * (double)v + mesh_normal(m, n[2]) * (double)w;
    }
    h.near = t.h.t; h.vnear = t.h.p;
    if (hits.size() == 1) { h.far = t.h.t; h.vfar = t.h.p; }                     // :384-408
    else { h.far = hits[1].h.t; h.vfar = hits[1].h.p; }                          // :411-436
    /* triangle_intersect_uvs, mesh.rs:61-77 */
    if (m.tri_uv != nullptr && m.n_texcoords != 0) {
        const uint32_t* c = &m.tri_uv[3 * t.i];
        const float* a = &m.texcoords[2 * c[0]];
        const float* b = &m.texcoords[2 * c[1]];
        const float* cc = &m.texcoords[2 * c[2]];
        float iu = t.h.u, iv = t.h.v, iw = 1.f - t.h.u - t.h.v;
        float uu = a[0] * iw + b[0] * iu + cc[0] * iv;
        float vv = a[1] * iw + b[1] * iu + cc[1] * iv;
        h.u = rem_euclid_f(uu, 1.f);
        h.v = rem_euclid_f(1.f - vv, 1.f);
    } else {
        h.u = 0.f; h.v = 0.f;
    }
    h.prim = t.i;
    return true;
}

/* ---- the scene ------------------------------------------------------------- */
struct Scene {
    const rt_scene_desc* d;
    uint64_t area_seed;
};

bool object_intersect(const Scene& sc, uint32_t oi, const Ray& ray, Hit& h) {
    const rt_object& o = sc.d->objects[oi];
    switch (o.kind) {
        case RT_OBJ_SPHERE: return sphere_intersect(sc.d->spheres[o.payload], ray, h);
        case RT_OBJ_PLANE: return plane_intersect(sc.d->planes[o.payload], ray, h);
        case RT_OBJ_BOX: {
            const rt_box& b = sc.d->boxes[o.payload];
            return box_intersect(v3(b.pos[0], b.pos[1], b.pos[2]), v3(b.half[0], b.half[1], b.half[2]), ray, h);   // aabb.rs:117-121
        }
        case RT_OBJ_MESH: return mesh_intersect(sc.d->meshes[o.payload], ray, h);
    }
    return false;
}

/* scene.rs:57-82: test every object in order, stable sort by near, take the first */
bool cast_ray_once(const Scene& sc, const Ray& ray, uint32_t& obj, Hit& hit) {
    struct E { uint32_t o; Hit h; };
    std::vector<E> v;
    for (uint32_t i = 0; i < sc.d->n_objects; i++) {
        Hit h;
        if (object_intersect(sc, i, ray, h)) v.push_back({i, h});
    }
    std::stable_sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.h.near < b.h.near; });
    if (v.empty()) return false;
    obj = v[0].o; hit = v[0].h;
    return true;
}

/* material.rs:119-141 */
Color texture_at(const rt_texture& t, float u, float v) {
    switch (t.kind) {
        case RT_TEX_SOLID: return {t.a[0], t.a[1], t.a[2]};
        case RT_TEX_IMAGE: {
            COUNT(texel_image);
            float w = (float)t.w, h = (float)t.h;
            uint32_t x = as_u32f(clampf(u * w, 0.f, w - 1.f));
            uint32_t y = as_u32f(clampf(v * h, 0.f, h - 1.f));
            const uint8_t* p = &t.rgb[3 * ((size_t)y * t.w + x)];
            return {p[0], p[1], p[2]};
        }
        case RT_TEX_CHECKER: {
            bool a = u > 0.5f, b = v > 0.5f;
            if (a == b) return {t.a[0], t.a[1], t.a[2]};
            return {t.b[0], t.b[1], t.b[2]};
        }
    }
    return {0, 0, 0};
}

/* Deliberate mistakes, for the tests' negative controls only (orc_set_mistake): a test that compares this file with a
 * render the reference's author made must FAIL when one of these is switched on, or it shows nothing.
 *   1  cubemap texels taken from the nearest texel instead of the four-tap blend of skybox.rs:70-84
 *   2  the x = -1 and x = +1 entries of the cubemap's face table (skybox.rs:96-99) exchanged */
std::atomic<uint32_t> g_mistake{0};

/* skybox.rs:64-86 */
Color poll_tex(const rt_skybox& sb, uint32_t cell, uint32_t cx, uint32_t cy, float xf, float yf) {
    double x = (double)(cx * cell) + (double)xf * (double)(cell - 1);
    double y = (double)(cy * cell) + (double)yf * (double)(cell - 1);
    if (g_mistake.load(std::memory_order_relaxed) & 1u) { x = std::floor(x + 0.5); y = std::floor(y + 0.5); }
    auto px = [&](double xx, double yy) -> Color {
        COUNT(texel_cubemap);
        uint32_t xi = as_u32(xx), yi = as_u32(yy);
        if (xi >= sb.w || yi >= sb.h) { COUNT(would_panic); return {0, 0, 0}; }   /* get_pixel would panic */
        const uint8_t* p = &sb.rgb[3 * ((size_t)yi * sb.w + xi)];
        return {p[0], p[1], p[2]};
    };
    if (std::fmod(x, 1.) == 0. && std::fmod(y, 1.) == 0.) return px(x, y);
    double xmin = std::floor(x), xmax = std::ceil(x);
    double ymin = std::floor(y), ymax = std::ceil(y);
    Color c00 = px(xmin, ymin), c10 = px(xmax, ymin), c01 = px(xmin, ymax), c11 = px(xmax, ymax);
    return to_color(blerp((x - xmin) / (xmax - xmin), (y - ymin) / (ymax - ymin), to_v3(c00), to_v3(c10), to_v3(c01), to_v3(c11)));
}

/* skybox.rs:15-31, 89-119 */
Color skybox_ray_color(const Scene& sc, const Ray& ray) {
    const rt_skybox& sb = sc.d->skybox;
    switch (sb.kind) {
        case RT_SKY_SOLID: return {sb.color[0], sb.color[1], sb.color[2]};
        case RT_SKY_NORMAL: return from_normal(ray.direction);
        case RT_SKY_CUBEMAP: {
            Ray r = ray_new(ray.direction * 2., -ray.direction);
            Hit h;
            Counts* save = tl_counts; tl_counts = nullptr;   /* the face-select box is not scene geometry */
            bool ok = box_intersect(v3(0., 0., 0.), v3(0.5, 0.5, 0.5), r, h);
            tl_counts = save;
            if (!ok) { COUNT(would_panic); return {0, 0, 0}; }                     /* unwrap() :92 */
            V3 normal = h.normal;
            uint32_t cx, cy;
            if (normal.z == -1.) { cx = 1; cy = 1; }
            else if (normal.x == -1.) { cx = 2; cy = 1; }
            else if (normal.x == 1.) { cx = 0; cy = 1; }
            else if (normal.z == 1.) { cx = 3; cy = 1; }
            else if (normal.y == 1.) { cx = 1; cy = 0; }
            else if (normal.y == -1.) { cx = 1; cy = 2; }
            else { COUNT(would_panic); return {0, 0, 0}; }                         /* unreachable!() :107 */
            if ((g_mistake.load(std::memory_order_relaxed) & 2u) && (normal.x == 1. || normal.x == -1.) && normal.z != -1.) cx = 2 - cx;
            float u = h.u, v = h.v;
            if (normal.y == 1.) { u = 1.f - h.u; v = 1.f - h.v; }
            else if (normal.y == -1.) { u = h.u; v = 1.f - h.v; }
            return poll_tex(sb, sb.w / 4, cx, cy, u, v);
        }
    }
    return {0, 0, 0};
}

/* lighting/mod.rs:16-31 */
struct LightShading { double diffuse = 0., specular = 0., intensity = 0.; };

/* area.rs:121 draws from an unseeded thread generator.  This stand-in makes area-light scenes runnable and
 * repeatable, but cannot match the reference's draws.  It is counter-based, so that a renderer which visits the rays
 * of a pixel in another order (level by level instead of depth first) draws the same numbers: every ray carries a
 * key -- a primary ray's from the seed and its pixel index, a child's from its parent's and its kind -- and the draws
 * of sample `it` of light `li` at the ray's hit are a splitmix64 sequence started from (key, li, it). */
inline uint64_t rng_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rng_key_primary(uint64_t seed, uint64_t pixel) { return rng_mix(seed ^ rng_mix(pixel)); }
enum : uint64_t { RNG_CHILD_TRANSMIT = 1, RNG_CHILD_REFLECT = 2 };
inline uint64_t rng_key_child(uint64_t key, uint64_t kind) { return rng_mix(key + kind); }
struct Rng {
    uint64_t s;
    static Rng for_sample(uint64_t key, uint32_t li, uint32_t it) { return Rng{rng_mix(key ^ rng_mix(((uint64_t)li << 32) | it))}; }
    double next_pm1() {   /* uniform in [-1, 1] */
        s += 0xD1342543DE82EF95ull;
        return (double)(rng_mix(s) >> 11) * (2.0 / 9007199254740991.0) - 1.0;
    }
};

LightShading point_like(const Scene& sc, const rt_light& L, V3 position, const Ray& ray, const Hit& hit, bool& too_far) {
    /* point.rs:66-105 (and the loop body of area.rs:117-159) */
    LightShading out;
    too_far = false;
    V3 lvec = position - hit.vnear;
    double dist = magnitude(lvec);
    if (dist > L.max_distance) { too_far = true; return out; }
    lvec = lvec / dist;
    double diffuse = clampd(dot(hit.normal, lvec), 0., 1.7976931348623157e308);
    V3 halfway_dir = normalize(lvec - ray.direction);
    double specular = powi(clampd(dot(hit.normal, halfway_dir), 0., 1.7976931348623157e308), L.specular_power);
    Ray shadow_ray = ray_new(hit.vnear + hit.normal * EPSILON, lvec);
    uint32_t so; Hit sh;
    COUNT(rays_shadow);
    CtxScope cs(CTX_SHADOW);
    if (cast_ray_once(sc, shadow_ray, so, sh)) {
        if (sh.near <= dist) { diffuse *= 0.; specular *= 0.; }
    }
    double q = dist / METER;
    out.diffuse = diffuse; out.specular = specular; out.intensity = L.intensity / (q * q);
    return out;
}

LightShading light_shading(const Scene& sc, const rt_light& L, uint32_t li, const Ray& ray, const Hit& hit, uint64_t key) {
    switch (L.kind) {
        case RT_LIGHT_SUN: {                                                       // sun.rs:72-97
            V3 lvec = -v3(L.v[0], L.v[1], L.v[2]);
            double diffuse = clampd(dot(hit.normal, lvec), 0., 1.7976931348623157e308);
            V3 halfway_dir = normalize(lvec - ray.direction);
            double specular = powi(clampd(dot(hit.normal, halfway_dir), 0., 1.7976931348623157e308), L.specular_power);
            if (L.shadows) {
                Ray shadow_ray = ray_new(hit.vnear + hit.normal * EPSILON, lvec);
                uint32_t so; Hit sh;
                COUNT(rays_shadow);
                CtxScope cs(CTX_SHADOW);
                if (cast_ray_once(sc, shadow_ray, so, sh)) {
                    diffuse *= L.shadow_coefficient;
                    specular *= L.shadow_coefficient;
                }
            }
            LightShading o; o.diffuse = diffuse; o.specular = specular; o.intensity = L.intensity;
            return o;
        }
        case RT_LIGHT_POINT: {
            bool far;
            return point_like(sc, L, v3(L.v[0], L.v[1], L.v[2]), ray, hit, far);
        }
        case RT_LIGHT_AREA: {                                                      // area.rs:114-166
            std::vector<LightShading> samples;
            for (uint32_t it = 0; it < L.area_iterations; it++) {
                Rng rng = Rng::for_sample(key, li, it);
                V3 pos;
                if (L.area_surface == RT_AREA_SPHERE) {                            // area.rs:30-40
                    double x = rng.next_pm1(), y = rng.next_pm1(), z = rng.next_pm1();
                    double mag = std::sqrt(x * x + y * y + z * z);
                    double d = (rng.next_pm1() * 0.5 + 0.5) / mag;
                    x *= d; y *= d; z *= d;
                    pos = v3(x, y, z) * L.area_radius + v3(L.v[0], L.v[1], L.v[2]);
                } else {                                                           // area.rs:41-48
                    const double (*c)[3] = L.area_corners;
                    double tx = rng.next_pm1() * 0.5 + 0.5;
                    double ty = rng.next_pm1() * 0.5 + 0.5;
                    pos = blerp(tx, ty, v3(c[0][0], c[0][1], c[0][2]), v3(c[1][0], c[1][1], c[1][2]),
                                v3(c[2][0], c[2][1], c[2][2]), v3(c[3][0], c[3][1], c[3][2]));
                }
                bool far;
                LightShading s = point_like(sc, L, pos, ray, hit, far);
                if (far) return LightShading();
                samples.push_back(s);
            }
            LightShading o;
            double sd = 0., ss = 0., si = 0.;
            for (auto& s : samples) sd += s.diffuse;
            for (auto& s : samples) ss += s.specular;
            for (auto& s : samples) si += s.intensity;
            double n = (double)samples.size();
            o.diffuse = sd / n; o.specular = ss / n; o.intensity = si / n;
            return o;
        }
    }
    return LightShading();
}

/* scene.rs:85-206 */
Color trace_ray(const Scene& sc, const Ray& ray, uint32_t depth, uint64_t key, uint32_t* id_out) {
    if (tl_counts && depth < 8) tl_counts->rays_per_level[depth]++;
    uint32_t oi; Hit hit;
    if (!cast_ray_once(sc, ray, oi, hit)) {
        if (id_out) { id_out[0] = RT_ID_NONE; id_out[1] = RT_ID_NONE; }
        return skybox_ray_color(sc, ray);
    }
    if (id_out) { id_out[0] = oi; id_out[1] = hit.prim; }
    const rt_object& object = sc.d->objects[oi];
    const rt_material& mat = sc.d->materials[object.material];

    V3 color = to_v3(texture_at(sc.d->textures[mat.texture], hit.u, hit.v));
    V3 base_color = color;
    if (mat.emissivity == 1.) return to_color(color);

    V3 sum_vecs = to_v3(Color{sc.d->options.ambient[0], sc.d->options.ambient[1], sc.d->options.ambient[2]});
    for (uint32_t li = 0; li < sc.d->n_lights; li++) {
        const rt_light& L = sc.d->lights[li];
        V3 lcol = to_v3(Color{L.color[0], L.color[1], L.color[2]});
        LightShading sh = light_shading(sc, L, li, ray, hit, key);
        V3 diffuse = lcol * sh.diffuse;
        V3 specular = lcol * (sh.specular * L.specular_strength);
        sum_vecs = sum_vecs + (diffuse + specular) * sh.intensity;
    }
    color = color * sum_vecs;

    double reflectiveness = mat.reflectiveness, transparency = mat.transparency, ior = mat.ior;
    const uint32_t max_depth = sc.d->options.max_ray_depth;
    if (transparency > EPSILON && depth < max_depth) {
        V3 transparency_color = color;
        if (ior == 1.) {
            COUNT(rays_transmit);
            Color thru = trace_ray(sc, ray_new(hit.vfar, ray.direction), depth + 1, rng_key_child(key, RNG_CHILD_TRANSMIT), nullptr);
            transparency_color = to_v3(thru);
        } else {
            V3 ref_vec;
            if (!refraction_vec(ray, hit.normal, 1., ior, ref_vec)) {
                COUNT(would_panic);                                               /* unwrap() :140 */
            } else {
                Hit ref_hit, exit_hit;
                bool ok1, ok2 = false;
                {
                    CtxScope cs(CTX_HELPER);
                    ok1 = object_intersect(sc, oi, ray_new(hit.vnear - ref_vec, ref_vec), ref_hit);
                    if (ok1) ok2 = object_intersect(sc, oi, ray_new(ref_hit.vfar + ref_vec, -ref_vec), exit_hit);
                }
                if (ok1) {
                    if (ok2) {
                        V3 exit_ref_vec;
                        if (refraction_vec(ray_new(ref_hit.vfar + ref_vec * EPSILON, ref_vec), -exit_hit.normal, ior, 1., exit_ref_vec)) {
                            COUNT(rays_transmit);
                            Color ref_col = trace_ray(sc, ray_new(ref_hit.vfar + exit_ref_vec * EPSILON, exit_ref_vec), depth + 1, rng_key_child(key, RNG_CHILD_TRANSMIT), nullptr);
                            transparency_color = to_v3(ref_col);
                        }
                    }
                }
            }
        }
        if (reflectiveness > EPSILON) {
            double d = dot(-ray.direction, hit.normal);
            d = d * d;                                                            /* powi(2) */
            COUNT(rays_reflect);
            Color reflected = trace_ray(sc, reflect(ray, hit.vnear + hit.normal * EPSILON, hit.normal), depth + 1, rng_key_child(key, RNG_CHILD_REFLECT), nullptr);
            transparency_color = vlerp(transparency_color, to_v3(reflected), 1. - d);
        }
        color = vlerp(color, transparency_color, transparency);
    }

    if (reflectiveness > EPSILON && depth < max_depth && transparency < EPSILON) {
        COUNT(rays_reflect);
        Color reflected = trace_ray(sc, reflect(ray, hit.vnear + hit.normal * EPSILON, hit.normal), depth + 1, rng_key_child(key, RNG_CHILD_REFLECT), nullptr);
        color = vlerp(color, to_v3(reflected), reflectiveness);
    }

    double emissivity = mat.emissivity;
    if (emissivity > 0.) return to_color(vlerp(color, base_color, emissivity));
    return to_color(color);
}

/* scene.rs:210-217 */
Color trace_pixel(const Scene& sc, int32_t x, int32_t y, uint64_t key, uint32_t* id_out) {
    Ray ray = ray_new(v3(sc.d->camera.origin[0], sc.d->camera.origin[1], sc.d->camera.origin[2]),
                      direction_at(sc.d->camera, (double)x, (double)y));
    COUNT(rays_primary);
    return trace_ray(sc, ray, 0, key, id_out);
}

int validate(const rt_scene_desc* d) {
    if (!d || d->camera.vw <= 0 || d->camera.vh <= 0) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        if (o.material >= d->n_materials) return RT_ERR_INVALID;
        uint32_t n = o.kind == RT_OBJ_SPHERE ? d->n_spheres : o.kind == RT_OBJ_PLANE ? d->n_planes
                   : o.kind == RT_OBJ_BOX ? d->n_boxes : o.kind == RT_OBJ_MESH ? d->n_meshes : 0;
        if (o.payload >= n) return RT_ERR_INVALID;
    }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->materials[i].texture >= d->n_textures) return RT_ERR_INVALID;
    return RT_OK;
}

void fill_stats(const Counts& c, rt_stats* s, orc_counts* oc) {
    if (s) {
        std::memset(s, 0, sizeof(*s));
        s->rays_primary = c.rays_primary; s->rays_reflect = c.rays_reflect;
        s->rays_transmit = c.rays_transmit; s->rays_shadow = c.rays_shadow;
        for (int i = 0; i < 8; i++) s->rays_per_level[i] = c.rays_per_level[i];
    }
    if (oc) {
        oc->node_tests = c.node_tests; oc->tri_tests = c.tri_tests; oc->sphere_tests = c.sphere_tests;
        oc->plane_tests = c.plane_tests; oc->box_tests = c.box_tests;
        oc->texel_image = c.texel_image; oc->texel_cubemap = c.texel_cubemap;
        oc->would_panic = c.would_panic;
        for (int x = 0; x < 3; x++) for (int k = 0; k < 5; k++) oc->by_ctx[x][k] = c.by_ctx[x][k];
    }
}

/* seed of the stand-in generator above; tests vary it to learn how far the draws move a pixel */
std::atomic<uint64_t> g_area_seed{0x9E3779B97F4A7C15ull};

}  // namespace

extern "C" {

void orc_set_area_seed(uint64_t seed) { g_area_seed.store(seed); }
void orc_set_mistake(uint32_t flags) { g_mistake.store(flags); }

/* scene.rs:220-246 over an explicit pixel list (pixel index i = y*vw + x, the
 * reference's own linear index); pixels == NULL renders 0..n in order. */
int rt_cpu_render_pixels(const rt_scene_desc* d, int n_threads, uint64_t n, const uint32_t* pixels,
                         uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out, orc_counts* counts_out) {
    int rc = validate(d);
    if (rc != RT_OK) return rc;
    if (!rgb_out) return RT_ERR_INVALID;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    const int32_t vw = d->camera.vw;
    const uint64_t total = (uint64_t)vw * (uint64_t)d->camera.vh;
    if (pixels) { for (uint64_t k = 0; k < n; k++) if (pixels[k] >= total) return RT_ERR_INVALID; }
    else if (n > total) return RT_ERR_INVALID;

    Scene sc{d, g_area_seed.load()};
    const uint64_t chunk = 256;
    std::atomic<uint64_t> next{0};
    std::vector<Counts> counts((size_t)n_threads);
    auto worker = [&](int tid) {
        tl_counts = &counts[(size_t)tid];
        for (;;) {
            uint64_t b = next.fetch_add(chunk);
            if (b >= n) break;
            uint64_t e = std::min(n, b + chunk);
            for (uint64_t k = b; k < e; k++) {
                uint64_t i = pixels ? pixels[k] : k;
                uint32_t ids[2];
                Color c = trace_pixel(sc, (int32_t)(i % (uint64_t)vw), (int32_t)(i / (uint64_t)vw), rng_key_primary(sc.area_seed, i), ids);   // scene.rs:241
                rgb_out[3 * k] = c.r; rgb_out[3 * k + 1] = c.g; rgb_out[3 * k + 2] = c.b;
                if (prim_id_out) { prim_id_out[2 * k] = ids[0]; prim_id_out[2 * k + 1] = ids[1]; }
            }
        }
        tl_counts = nullptr;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    Counts sum;
    for (auto& c : counts) sum.add(c);
    fill_stats(sum, stats_out, counts_out);
    return RT_OK;
}

int rt_cpu_render(const rt_scene_desc* d, int n_threads, uint8_t* rgb_out, uint32_t* prim_id_out,
                  rt_stats* stats_out, orc_counts* counts_out) {
    if (!d) return RT_ERR_INVALID;
    return rt_cpu_render_pixels(d, n_threads, (uint64_t)d->camera.vw * (uint64_t)d->camera.vh, nullptr,
                                rgb_out, prim_id_out, stats_out, counts_out);
}

/* Scene::cast_ray_once (scene.rs:79-82) for arbitrary rays */
int rt_cpu_cast_rays(const rt_scene_desc* d, uint32_t n, const double* origins, const double* dirs, rt_ray_hit* out) {
    int rc = validate(d);
    if (rc != RT_OK) return rc;
    Scene sc{d, 0};
    for (uint32_t k = 0; k < n; k++) {
        Ray r = ray_new(v3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]), v3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
        uint32_t oi; Hit h;
        rt_ray_hit& o = out[k];
        std::memset(&o, 0, sizeof(o));
        if (cast_ray_once(sc, r, oi, h)) {
            o.object = oi; o.prim = h.prim; o.near_t = h.near; o.far_t = h.far;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.vnear[0] = h.vnear.x; o.vnear[1] = h.vnear.y; o.vnear[2] = h.vnear.z;
            o.vfar[0] = h.vfar.x; o.vfar[1] = h.vfar.y; o.vfar[2] = h.vfar.z;
            o.uv[0] = h.u; o.uv[1] = h.v;
        } else {
            o.object = RT_ID_NONE; o.prim = RT_ID_NONE;
        }
    }
    return RT_OK;
}

/* ---- primitive-level entry points for the known-answer tests ---------------- */
static void put_hit(const Hit& h, bool ok, rt_ray_hit* o) {
    std::memset(o, 0, sizeof(*o));
    o->object = ok ? 0u : RT_ID_NONE; o->prim = ok ? h.prim : RT_ID_NONE;
    if (!ok) return;
    o->near_t = h.near; o->far_t = h.far;
    o->normal[0] = h.normal.x; o->normal[1] = h.normal.y; o->normal[2] = h.normal.z;
    o->vnear[0] = h.vnear.x; o->vnear[1] = h.vnear.y; o->vnear[2] = h.vnear.z;
    o->vfar[0] = h.vfar.x; o->vfar[1] = h.vfar.y; o->vfar[2] = h.vfar.z;
    o->uv[0] = h.u; o->uv[1] = h.v;
}
static Ray mkray(const double* o, const double* d) { return ray_new(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])); }

int orc_sphere_intersect(const rt_sphere* s, const double* o, const double* d, rt_ray_hit* out) {
    Hit h; bool ok = sphere_intersect(*s, mkray(o, d), h); put_hit(h, ok, out); return ok;
}
int orc_plane_intersect(const rt_plane* p, const double* o, const double* d, rt_ray_hit* out) {
    Hit h; bool ok = plane_intersect(*p, mkray(o, d), h); put_hit(h, ok, out); return ok;
}
int orc_box_intersect(const rt_box* b, const double* o, const double* d, rt_ray_hit* out) {
    Hit h; bool ok = box_intersect(v3(b->pos[0], b->pos[1], b->pos[2]), v3(b->half[0], b->half[1], b->half[2]), mkray(o, d), h);
    put_hit(h, ok, out); return ok;
}
int orc_aabb_bool(const double* mn, const double* mx, const double* o, const double* d) {
    rt_bvh_node nd; for (int i = 0; i < 3; i++) { nd.min[i] = mn[i]; nd.max[i] = mx[i]; } nd.count = 0; nd.right_or_first = 0;
    return aabb_bool(nd, mkray(o, d));
}
/* out: t, u, v, px, py, pz */
int orc_triangle_intersect(const double* v0, const double* v1, const double* v2, const double* o, const double* d, double* out) {
    TriIntersect ti;
    bool ok = triangle_intersect(v3(v0[0], v0[1], v0[2]), v3(v1[0], v1[1], v1[2]), v3(v2[0], v2[1], v2[2]), mkray(o, d), ti);
    if (ok) { out[0] = ti.t; out[1] = ti.u; out[2] = ti.v; out[3] = ti.p.x; out[4] = ti.p.y; out[5] = ti.p.z; }
    return ok;
}
int orc_refraction_vec(const double* d, const double* n, double from_ior, double to_ior, double* out) {
    V3 r; Ray ray = ray_new(v3(0, 0, 0), v3(d[0], d[1], d[2]));
    bool ok = refraction_vec(ray, v3(n[0], n[1], n[2]), from_ior, to_ior, r);
    if (ok) { out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    return ok;
}
void orc_color_newf(double r, double g, double b, uint8_t* out) { Color c = newf(r, g, b); out[0] = c.r; out[1] = c.g; out[2] = c.b; }
void orc_camera_direction_at(const rt_camera* cam, double x, double y, double* out) {
    V3 d = direction_at(*cam, x, y); out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
double orc_camera_chf(double fov) { return std::tan((90. - fov * 0.5) * 0.017453); }   /* camera.rs:45-47 */
void orc_texture_at(const rt_texture* t, float u, float v, uint8_t* out) { Color c = texture_at(*t, u, v); out[0] = c.r; out[1] = c.g; out[2] = c.b; }
void orc_skybox_ray_color(const rt_skybox* sb, const double* d, uint8_t* out) {
    rt_scene_desc desc; std::memset(&desc, 0, sizeof(desc)); desc.skybox = *sb;
    Scene sc{&desc, 0};
    Color c = skybox_ray_color(sc, ray_new(v3(0, 0, 0), v3(d[0], d[1], d[2])));
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
double orc_powi(double a, int b) { return powi(a, b); }

}  // extern "C"
