// Device functions of the render path: ray/primitive intersection, BVH
// traversal, texture / skybox sampling and light shading, each following the
// cited reference lines operation by operation (f64, unfused: this file is
// compiled with -ffp-contract=off so no multiply-add is contracted).
#pragma once
#include <hip/hip_runtime.h>

#include <hip/hip_fp16.h>
#include "dev_types.h"

#define DEV __device__ __forceinline__

// Pointers read out of device structs are generic to the compiler; casting them to the global address space
// makes the loads global_load (one counter to wait on) instead of flat_load.
#define RT_GLOBAL __attribute__((address_space(1)))
template <typename T> DEV const RT_GLOBAL T* gptr(const T* p) { return (const RT_GLOBAL T*)p; }

// ---------------------------------------------------------------- vector.rs
DEV D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV D3 operator*(D3 a, D3 b) { return d3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
DEV D3 operator/(D3 a, D3 b) { return d3(a.x / b.x, a.y / b.y, a.z / b.z); }
DEV D3 operator/(D3 a, double s) { return d3(a.x / s, a.y / s, a.z / s); }
DEV D3 operator-(D3 a) { return d3(-a.x, -a.y, -a.z); }
DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // vector.rs:60-62
DEV D3 cross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x); }  // :65-71
DEV double magnitude(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }                 // :74-76
// a / s for s = magnitude(a) (vector.rs:192-202: three divisions by one divisor).  The compiler's f64 division is
// v_div_scale on both operands, v_rcp_f64 and two Newton steps on the reciprocal, q0 = n * r, rem = fma(-d, q0, n),
// q = fma(rem, r, q0), v_div_fixup.  Scale and fixup change nothing while divisor and quotient stay clear of the ends of
// the exponent range and no operand is 0, inf or NaN, and the reciprocal's part depends on the divisor alone: here it is
// worked out once for the three numerators, by the same operations in the same order, so the quotients are the same
// bits (14 instructions instead of 39; tools/div_check.hip compares the two over the exponent range).  Outside that
// range -- a component below 2^-200, zeros included, or a magnitude outside [2^-200, 2^200] -- the plain divisions run.
DEV D3 div_by_norm(D3 a, double s) {
    const double LO = 0x1p-200, HI = 0x1p200;
    if (s >= LO && s <= HI && fabs(a.x) >= LO && fabs(a.y) >= LO && fabs(a.z) >= LO) {
        double r = __builtin_amdgcn_rcp(s);
        r = fma(r, fma(-s, r, 1.0), r);
        r = fma(r, fma(-s, r, 1.0), r);
        double qx = a.x * r, qy = a.y * r, qz = a.z * r;
        qx = fma(fma(-s, qx, a.x), r, qx);
        qy = fma(fma(-s, qy, a.y), r, qy);
        qz = fma(fma(-s, qz, a.z), r, qz);
        return d3(qx, qy, qz);
    }
    return a / s;
}
DEV D3 normalize(D3 a) { return div_by_norm(a, magnitude(a)); }                                // :79-81
template <typename P> DEV D3 ld3(P p) { return d3(p[0], p[1], p[2]); }

// Rust f64::signum: NaN -> NaN, otherwise +-1 with the sign of x (signum(+-0) = +-1)
DEV double signum(double x) { return x != x ? x : copysign(1.0, x); }
// Rust f64::clamp: NaN stays NaN
DEV double clampd(double x, double lo, double hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
// a distance limit as f32, not below the f64 value: rounded to nearest, then moved up by more than that rounding
DEV float limit_up(double b) {
    const float f = (float)b;
    return fmaf(fabsf(f), 1.2e-7f, f) + 1.1754944e-38f;
}
DEV float clampf(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
// Rust `as u8` / `as u32`: truncation toward zero, saturating, NaN -> 0
DEV uint32_t as_u8(double v) { if (!(v > 0.0)) return 0u; if (v >= 255.0) return 255u; return (uint32_t)(int)v; }
DEV uint32_t as_u32d(double v) { if (!(v > 0.0)) return 0u; if (v >= 4294967295.0) return 4294967295u; return (uint32_t)v; }
DEV uint32_t as_u32f(float v) { if (!(v > 0.0f)) return 0u; if (v >= 4294967296.0f) return 4294967295u; return (uint32_t)v; }
DEV float rem_euclid_f(float a, float b) { float r = fmodf(a, b); return r < 0.0f ? r + fabsf(b) : r; }
// f64::powi with a run-time exponent = compiler-rt __powidf2
DEV double rt_powi(double a, int b) {
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
DEV double lerp(double a, double b, double c) { return a + (b - a) * c; }                      // math/mod.rs:17-19
DEV D3 vlerp(D3 a, D3 b, double t) { return d3(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }

// Sample points of area lights.  area.rs:121 draws from an unseeded thread generator; here the draws are a function of
// (seed, pixel, the ray's place in the pixel's tree of rays, light, sample), so that they do not depend on the order
// in which rays are traced: every ray carries a key -- a primary ray's from the seed and its pixel index, a child's
// from its parent's and its kind -- and the draws of sample `it` of light `li` at the ray's hit are a splitmix64
// sequence started from (key, li, it).
DEV unsigned long long rng_mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
DEV unsigned long long rng_key_primary(unsigned long long seed, unsigned long long pixel) { return rng_mix(seed ^ rng_mix(pixel)); }
#define RT_RNG_CHILD_TRANSMIT 1ull
#define RT_RNG_CHILD_REFLECT 2ull
DEV unsigned long long rng_key_child(unsigned long long key, unsigned long long kind) { return rng_mix(key + kind); }
struct AreaRng {
    unsigned long long s;
    DEV double next_pm1() {   // uniform in [-1, 1]
        s += 0xD1342543DE82EF95ull;
        return (double)(rng_mix(s) >> 11) * (2.0 / 9007199254740991.0) - 1.0;
    }
};
// AreaSurface::sample (area.rs:25-50) for sample `it` of light `li` at the hit of the ray with this key
template <typename P>
DEV D3 area_sample(P Lt, unsigned long long key) {
    AreaRng rng = {rng_mix(key ^ rng_mix(((unsigned long long)Lt->area_li << 32) | Lt->area_it))};
    if (Lt->area_surface == 0) {                                                    // Sphere, area.rs:30-40
        double x = rng.next_pm1(), y = rng.next_pm1(), z = rng.next_pm1();
        const double mag = sqrt(x * x + y * y + z * z);
        const double d = (rng.next_pm1() * 0.5 + 0.5) / mag;
        x *= d; y *= d; z *= d;
        return d3(x, y, z) * Lt->area_radius + d3(Lt->v[0], Lt->v[1], Lt->v[2]);
    }
    const double tx = rng.next_pm1() * 0.5 + 0.5;                                    // Rectangle, area.rs:41-48; blerp: math/mod.rs:22-26
    const double ty = rng.next_pm1() * 0.5 + 0.5;
    const D3 c00 = d3(Lt->area_corners[0][0], Lt->area_corners[0][1], Lt->area_corners[0][2]);
    const D3 c10 = d3(Lt->area_corners[1][0], Lt->area_corners[1][1], Lt->area_corners[1][2]);
    const D3 c01 = d3(Lt->area_corners[2][0], Lt->area_corners[2][1], Lt->area_corners[2][2]);
    const D3 c11 = d3(Lt->area_corners[3][0], Lt->area_corners[3][1], Lt->area_corners[3][2]);
    const D3 a = c00 * (1. - tx) + c10 * tx;
    const D3 b = c01 * (1. - tx) + c11 * tx;
    return a * (1. - ty) + b * ty;
}

// material.rs:38-44 (Color::newf) packed r | g<<8 | b<<16
DEV uint32_t newf(double r, double g, double b) {
    return as_u8(clampd(r, 0., 1.) * 255.0) | (as_u8(clampd(g, 0., 1.) * 255.0) << 8) | (as_u8(clampd(b, 0., 1.) * 255.0) << 16);
}
DEV uint32_t to_color(D3 v) { return newf(v.x, v.y, v.z); }
DEV D3 to_d3(uint32_t c) { return d3((double)(c & 255u) / 255., (double)((c >> 8) & 255u) / 255., (double)((c >> 16) & 255u) / 255.); }  // vector.rs:216-224

// ---------------------------------------------------------------- ray.rs
struct Ray {
    D3 o, d, m;          // origin, direction, 1/direction (ray.rs:17-23)
    // f32 view for BVH culling: origin as an unevaluated sum hi + lo (so that no origin
    // rounding enters the slab distances), 1/direction rounded to f32, direction signs
    float ohx, ohy, ohz, olx, oly, olz, mfx, mfy, mfz;
    bool nx, ny, nz;     // sign bit of the direction: the near slab bound is max when set (s = -signum(d), aabb.rs:21-25)
};
DEV Ray ray_new(D3 o, D3 d) {
    Ray r; r.o = o; r.d = d; r.m = d3(1. / d.x, 1. / d.y, 1. / d.z);
    r.ohx = (float)o.x; r.olx = (float)(o.x - (double)r.ohx);
    r.ohy = (float)o.y; r.oly = (float)(o.y - (double)r.ohy);
    r.ohz = (float)o.z; r.olz = (float)(o.z - (double)r.ohz);
    r.mfx = (float)r.m.x; r.mfy = (float)r.m.y; r.mfz = (float)r.m.z;
    r.nx = __double2hiint(d.x) < 0; r.ny = __double2hiint(d.y) < 0; r.nz = __double2hiint(d.z) < 0;
    return r;
}
DEV D3 along(const Ray& r, double t) { return r.o + r.d * t; }                                 // ray.rs:26-28

// math/mod.rs:48-63
DEV bool refraction_vec(D3 dir, D3 normal, double from_ior, double to_ior, D3& out) {
    double n = from_ior / to_ior;
    double cos_i = -dot(normal, dir);
    double sin_t2 = n * n * (1. - cos_i * cos_i);
    if (sin_t2 > 1.) return false;
    double cos_t = sqrt(1. - sin_t2);
    out = dir * n + normal * (n * cos_i - cos_t);
    return true;
}

// ---------------------------------------------------------------- object/mod.rs:18-36
struct Hit { D3 normal; double near; D3 vnear; double far; D3 vfar; float u, v; uint32_t prim; };

// sphere.rs:33-62.  `near_only`: just the hit test and `near`.
template <typename P> DEV bool sphere_near(P s, const Ray& ray, double& near) {
    D3 origin = ld3(s);
    double rad2 = s[3] * s[3];
    D3 l = origin - ray.o;
    double t2 = dot(l, ray.d);
    if (t2 < 0.0) return false;
    double d2 = dot(l, l) - t2 * t2;
    if (d2 > rad2) return false;
    double t3 = sqrt(rad2 - d2);
    near = t2 - t3;
    return true;
}
template <typename P> DEV bool sphere_intersect(P s, const Ray& ray, Hit& h) {
    D3 origin = ld3(s);
    double rad2 = s[3] * s[3];
    D3 l = origin - ray.o;
    double t2 = dot(l, ray.d);
    if (t2 < 0.0) return false;
    double d2 = dot(l, l) - t2 * t2;
    if (d2 > rad2) return false;
    double t3 = sqrt(rad2 - d2);
    double t0 = t2 - t3;
    double t1 = t2 + t3;
    D3 vtn = along(ray, t0);
    D3 vtf = along(ray, t1);
    D3 norm = normalize(vtn - origin);
    const float PI = 3.14159265358979323846f;
    h.normal = norm; h.near = t0; h.vnear = vtn; h.far = t1; h.vfar = vtf;
    h.u = 0.5f + (float)atan2(norm.x, norm.z) / (PI * 2.f);
    h.v = 0.5f - (float)asin(norm.y) / PI;
    h.prim = RT_NONE;
    return true;
}

// plane.rs:49-79
template <typename P> DEV bool plane_near(P p, const Ray& ray, double& near) {
    D3 origin = ld3(p), normal = ld3(p + 3);
    double denom = dot(normal, ray.d);
    if (fabs(denom) > RT_EPSILON) {
        double t = dot(origin - ray.o, normal) / denom;
        if (t > 0.) { near = t; return true; }
    }
    return false;
}
template <typename P> DEV bool plane_intersect(P p, const Ray& ray, Hit& h) {
    D3 origin = ld3(p), normal = ld3(p + 3);
    float uv_wrap = (float)p[6];
    double denom = dot(normal, ray.d);
    if (fabs(denom) > RT_EPSILON) {
        double t = dot(origin - ray.o, normal) / denom;
        if (t > 0.) {
            D3 pt = along(ray, t);
            h.normal = normal * -signum(denom);
            h.near = t; h.vnear = pt; h.far = t; h.vfar = pt;
            if (normal.y != 0.) {
                h.u = rem_euclid_f((float)pt.x / uv_wrap, 1.f);
                h.v = rem_euclid_f((float)pt.z / uv_wrap, 1.f);
            } else { h.u = 0.f; h.v = 0.f; }
            h.prim = RT_NONE;
            return true;
        }
    }
    return false;
}

// aabb.rs:18-67
DEV bool box_slabs(D3 pos, D3 size, const Ray& ray, D3& s, D3& t1, double& tn, double& tf) {
    D3 ro = ray.o - pos;
    s = d3(-signum(ray.d.x), -signum(ray.d.y), -signum(ray.d.z));
    t1 = ray.m * (-ro + (s * size));
    D3 t2 = ray.m * (-ro - (s * size));
    tn = fmax(fmax(t1.x, t1.y), t1.z);
    tf = fmin(fmin(t2.x, t2.y), t2.z);
    return !(tn > tf || tf < 0.);
}
template <typename P> DEV bool box_near(P b, const Ray& ray, double& near) {
    D3 s, t1; double tf;
    return box_slabs(ld3(b), ld3(b + 3), ray, s, t1, near, tf);
}
DEV bool box_intersect(D3 pos, D3 size, const Ray& ray, Hit& h) {
    D3 s, t1; double tn, tf;
    if (!box_slabs(pos, size, ray, s, t1, tn, tf)) return false;
    D3 normal;
    if (t1.x > t1.y && t1.x > t1.z) normal = d3(s.x, 0., 0.);
    else if (t1.y > t1.z) normal = d3(0., s.y, 0.);
    else normal = d3(0., 0., s.z);
    D3 pn = along(ray, tn);
    D3 pf = along(ray, tf);
    D3 an = d3(fabs(normal.x), fabs(normal.y), fabs(normal.z));
    D3 pns = (d3(1., 1., 1.) - an) * (pn - pos) / size;
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
    h.prim = RT_NONE;
    return true;
}

// ---------------------------------------------------------------- mesh BVH
// The slab test of acceleration.rs:134-151 in f32 on an outward-rounded box, made conservative: the entry
// distance is pushed down and the exit distance up by 2^-20 relative, which covers the rounding of the two
// subtractions, of 1/direction and of the product (<= 6 * 2^-24).  NaN slabs (0 * inf) are ignored by
// fmaxf/fminf exactly as f64::max/min ignore them in the reference.  (ax..az: the bounds the ray meets
// first on each axis, bx..bz: the ones it meets last.)
#define RT_CULL_SLACK 9.5367431640625e-07f
DEV bool slab_test2(float ax, float ay, float az, float bx, float by, float bz, const Ray& ray, float& tn_out, float& tf_out) {
    const float t1x = ((ax - ray.ohx) - ray.olx) * ray.mfx, t2x = ((bx - ray.ohx) - ray.olx) * ray.mfx;
    const float t1y = ((ay - ray.ohy) - ray.oly) * ray.mfy, t2y = ((by - ray.ohy) - ray.oly) * ray.mfy;
    const float t1z = ((az - ray.ohz) - ray.olz) * ray.mfz, t2z = ((bz - ray.ohz) - ray.olz) * ray.mfz;
    float tn = fmaxf(fmaxf(t1x, t1y), t1z);
    float tf = fminf(fminf(t2x, t2y), t2z);
    tn = tn - fabsf(tn) * RT_CULL_SLACK;
    tf = tf + fabsf(tf) * RT_CULL_SLACK;
    tn_out = tn; tf_out = tf;
    return !(tn > tf || tf < 0.f);
}
DEV bool slab_test(float ax, float ay, float az, float bx, float by, float bz, const Ray& ray, float& tn_out) {
    float tf;
    return slab_test2(ax, ay, az, bx, by, bz, ray, tn_out, tf);
}
template <typename FP>
DEV bool node_box2(FP bmin, FP bmax, const Ray& ray, float& tn_out, float& tf_out) {
    return slab_test2(ray.nx ? bmax[0] : bmin[0], ray.ny ? bmax[1] : bmin[1], ray.nz ? bmax[2] : bmin[2],
                      ray.nx ? bmin[0] : bmax[0], ray.ny ? bmin[1] : bmax[1], ray.nz ? bmin[2] : bmax[2], ray, tn_out, tf_out);
}
template <typename FP>
DEV bool node_box(FP bmin, FP bmax, const Ray& ray, float& tn_out) {
    float tf;
    return node_box2(bmin, bmax, ray, tn_out, tf);
}

// Per-lane constants of the packet walk's box test.  The slab distance (b - o) * m is evaluated as
// fma(b, m, c) with c = -(o * m) rounded to f32: one instruction per plane (two planes with the packed form), one
// rounding of the result.  Against the exact distance the value is off by at most
//     2^-24 |o m|  (rounding of c)  +  2^-24 |b m|  (rounding of m)  +  2^-24 |t|  (rounding of the result),
// and |t| <= (|b| + |o|) |m|, so E = 2^-22 (|o| + B) |m| with B = the largest |coordinate| of the mesh bounds all
// three.  c is moved outward by E per plane (entry planes down, exit planes up), after which the comparisons need no
// further slack.  1/direction is clamped to +-1e30 so that an axis-parallel ray gives no inf - inf.
typedef float __attribute__((ext_vector_type(2))) rt_f2;
struct PRay { float mx, my, mz, cminx, cminy, cminz, cmaxx, cmaxy, cmaxz; };
DEV void pray_axis(double o, double m, float B, float& mf, float& cmin, float& cmax) {
    mf = fminf(fmaxf((float)m, -1e30f), 1e30f);
    const double om = o * (double)mf;
    const float slack = (fabsf((float)o) + B) * fabsf(mf) * 2.4e-7f;   // 2^-22, rounded up
    const float lo = __double2float_rd(-om) - slack, hi = __double2float_ru(-om) + slack;
    const bool pos = !(mf < 0.f);
    cmin = pos ? lo : hi;   // the min plane is the entry plane when the ray runs in +axis direction
    cmax = pos ? hi : lo;
}
// The same from the direction itself, without the f64 division: 1/direction as the f32 reciprocal (1 ulp) of the
// direction rounded to f32, so the slope is off by up to 3 * 2^-24 instead of 2^-24 and the total by 5 * 2^-24
// (|o| + B) |m|; c is rounded to nearest (that rounding is the first term above) and c -+ slack rounds once more,
// 6 * 2^-24 in all: the slack is 9 * 2^-24.  (Directed roundings of c would be a dozen instructions each here, and
// these constants are worked out again after every round of triangle tests.)
DEV void pray_axis_d(double o, double d, float B, float& mf, float& cmin, float& cmax) {
    mf = fminf(fmaxf(__builtin_amdgcn_rcpf((float)d), -1e30f), 1e30f);
    const float c = (float)-(o * (double)mf);
    const float slack = (fabsf((float)o) + B) * fabsf(mf) * 5.4e-7f;   // 9 * 2^-24, rounded up
    const float lo = c - slack, hi = c + slack;
    const bool pos = !(mf < 0.f);
    cmin = pos ? lo : hi;
    cmax = pos ? hi : lo;
}
DEV PRay pray_new_d(const D3& o, const D3& d, float B) {
    PRay p;
    pray_axis_d(o.x, d.x, B, p.mx, p.cminx, p.cmaxx);
    pray_axis_d(o.y, d.y, B, p.my, p.cminy, p.cmaxy);
    pray_axis_d(o.z, d.z, B, p.mz, p.cminz, p.cmaxz);
    return p;
}
DEV PRay pray_new(const Ray& r, float B) {
    PRay p;
    pray_axis(r.o.x, r.m.x, B, p.mx, p.cminx, p.cmaxx);
    pray_axis(r.o.y, r.m.y, B, p.my, p.cminy, p.cmaxy);
    pray_axis(r.o.z, r.m.z, B, p.mz, p.cminz, p.cmaxz);
    return p;
}
// children k and k+1 of a uniform node against this lane's ray: entry and exit distance of each
#define RT_CHILD_PAIR(n, k, tn_a, tf_a, tn_b, tf_b)                                                                        \
    do {                                                                                                                  \
        const rt_f2 lx_ = __builtin_elementwise_fma(rt_f2{asf((n).minx[k]), asf((n).minx[k + 1])}, rt_f2{pr.mx, pr.mx}, rt_f2{pr.cminx, pr.cminx}); \
        const rt_f2 hx_ = __builtin_elementwise_fma(rt_f2{asf((n).maxx[k]), asf((n).maxx[k + 1])}, rt_f2{pr.mx, pr.mx}, rt_f2{pr.cmaxx, pr.cmaxx}); \
        const rt_f2 ly_ = __builtin_elementwise_fma(rt_f2{asf((n).miny[k]), asf((n).miny[k + 1])}, rt_f2{pr.my, pr.my}, rt_f2{pr.cminy, pr.cminy}); \
        const rt_f2 hy_ = __builtin_elementwise_fma(rt_f2{asf((n).maxy[k]), asf((n).maxy[k + 1])}, rt_f2{pr.my, pr.my}, rt_f2{pr.cmaxy, pr.cmaxy}); \
        const rt_f2 lz_ = __builtin_elementwise_fma(rt_f2{asf((n).minz[k]), asf((n).minz[k + 1])}, rt_f2{pr.mz, pr.mz}, rt_f2{pr.cminz, pr.cminz}); \
        const rt_f2 hz_ = __builtin_elementwise_fma(rt_f2{asf((n).maxz[k]), asf((n).maxz[k + 1])}, rt_f2{pr.mz, pr.mz}, rt_f2{pr.cmaxz, pr.cmaxz}); \
        (tn_a) = fmaxf(fmaxf(fminf(lx_.x, hx_.x), fminf(ly_.x, hy_.x)), fminf(lz_.x, hz_.x));                              \
        (tf_a) = fminf(fminf(fmaxf(lx_.x, hx_.x), fmaxf(ly_.x, hy_.x)), fmaxf(lz_.x, hz_.x));                              \
        (tn_b) = fmaxf(fmaxf(fminf(lx_.y, hx_.y), fminf(ly_.y, hy_.y)), fminf(lz_.y, hz_.y));                              \
        (tf_b) = fminf(fminf(fmaxf(lx_.y, hx_.y), fmaxf(ly_.y, hy_.y)), fmaxf(lz_.y, hz_.y));                              \
    } while (0)
// the same for a record whose planes are float arrays (a lane's own record, fetched by vector loads)
#define RT_CHILD_PAIR_F(n, k, tn_a, tf_a, tn_b, tf_b)                                                                      \
    do {                                                                                                                  \
        const rt_f2 lx_ = __builtin_elementwise_fma(rt_f2{(n).minx[k], (n).minx[k + 1]}, rt_f2{pr.mx, pr.mx}, rt_f2{pr.cminx, pr.cminx}); \
        const rt_f2 hx_ = __builtin_elementwise_fma(rt_f2{(n).maxx[k], (n).maxx[k + 1]}, rt_f2{pr.mx, pr.mx}, rt_f2{pr.cmaxx, pr.cmaxx}); \
        const rt_f2 ly_ = __builtin_elementwise_fma(rt_f2{(n).miny[k], (n).miny[k + 1]}, rt_f2{pr.my, pr.my}, rt_f2{pr.cminy, pr.cminy}); \
        const rt_f2 hy_ = __builtin_elementwise_fma(rt_f2{(n).maxy[k], (n).maxy[k + 1]}, rt_f2{pr.my, pr.my}, rt_f2{pr.cmaxy, pr.cmaxy}); \
        const rt_f2 lz_ = __builtin_elementwise_fma(rt_f2{(n).minz[k], (n).minz[k + 1]}, rt_f2{pr.mz, pr.mz}, rt_f2{pr.cminz, pr.cminz}); \
        const rt_f2 hz_ = __builtin_elementwise_fma(rt_f2{(n).maxz[k], (n).maxz[k + 1]}, rt_f2{pr.mz, pr.mz}, rt_f2{pr.cmaxz, pr.cmaxz}); \
        (tn_a) = fmaxf(fmaxf(fminf(lx_.x, hx_.x), fminf(ly_.x, hy_.x)), fminf(lz_.x, hz_.x));                              \
        (tf_a) = fminf(fminf(fmaxf(lx_.x, hx_.x), fmaxf(ly_.x, hy_.x)), fmaxf(lz_.x, hz_.x));                              \
        (tn_b) = fmaxf(fmaxf(fminf(lx_.y, hx_.y), fminf(ly_.y, hy_.y)), fminf(lz_.y, hz_.y));                              \
        (tf_b) = fminf(fminf(fmaxf(lx_.y, hx_.y), fmaxf(ly_.y, hy_.y)), fmaxf(lz_.y, hz_.y));                              \
    } while (0)

// mesh.rs:25-59 with the edges precomputed.  The 80-byte record is fetched as five 16-byte loads (one L1 tag
// lookup each) rather than field by field.
typedef double __attribute__((ext_vector_type(2))) rt_d2;
struct TriData { D3 v0, e1, e2; uint32_t orig, last; };
template <typename TP>
DEV TriData tri_load(TP T) {
    const RT_GLOBAL rt_d2* p = (const RT_GLOBAL rt_d2*)T;
    const rt_d2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriData r;
    r.v0 = d3(a.x, a.y, b.x); r.e1 = d3(b.y, c.x, c.y); r.e2 = d3(d.x, d.y, e.x);
    const unsigned long long w = (unsigned long long)__double_as_longlong(e.y);
    r.orig = (uint32_t)w; r.last = (uint32_t)(w >> 32);
    return r;
}
DEV bool tri_test_od(const TriData& T, const D3& ro, const D3& rd, double& t_out, float& u_out, float& v_out) {
    const D3 v0 = T.v0, edge1 = T.e1, edge2 = T.e2;
    D3 h = cross(rd, edge2);
    double a = dot(edge1, h);
    if (a > -RT_EPSILON && a < RT_EPSILON) return false;
    double f = 1. / a;
    D3 s = ro - v0;
    double u = f * dot(s, h);
    if (!(u >= 0.0 && u <= 1.0)) return false;
    D3 q = cross(s, edge1);
    double v = f * dot(rd, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(edge2, q);
    if (t > RT_EPSILON) { t_out = t; u_out = (float)u; v_out = (float)v; return true; }
    return false;
}
DEV bool tri_test(const TriData& T, const Ray& ray, double& t_out, float& u_out, float& v_out) {
    return tri_test_od(T, ray.o, ray.d, t_out, u_out, v_out);
}

struct MeshHit { double t0, t1; uint32_t r0; float u, v; uint32_t n; };   // n = hits kept (0, 1, 2)

// Per-lane traversal context: the lane's stack column in LDS, and the work the traversal really did
// (inner-node visits = two box tests each, triangle tests); only the instrumented kernel variants store the
// counts, elsewhere the increments are dead code.
struct Visits { uint32_t nodes, tris; uint32_t wnodes, wtris; uint32_t units; };   // units: 16-byte words of BVH data this lane fetched (a packet's shared fetches are booked on lane 0)   // per-lane counts; wave-level counts of the packet walk (same in every lane)

// Traversal stack: (node reference, entry distance) pairs.  The first RT_LDS_STACK entries of every lane live
// in LDS, laid out [entry][lane] so that a wave's push or pop touches consecutive 8-byte words (no bank
// conflicts); deeper entries spill to private memory.
#ifndef RT_BLOCK_DIM
#define RT_BLOCK_DIM 256
#endif
__shared__ uint2 rt_lds_stack[RT_LDS_STACK * RT_BLOCK_DIM];
// push/pop on plain locals (sp, spill): kept out of any struct so that the stack pointer stays in a register.
// The spill array is volatile so that the compiler keeps its (cold) accesses apart from the LDS ones instead
// of folding both into one flat access through a selected generic pointer.
#define RT_STACK_PUSH(ref_, tn_)                                                                          \
    do {                                                                                                  \
        if (sp < RT_LDS_STACK) rt_lds_stack[sp * RT_BLOCK_DIM + threadIdx.x] = make_uint2((ref_), __float_as_uint(tn_)); \
        else spill[sp - RT_LDS_STACK] = (unsigned long long)(ref_) | ((unsigned long long)__float_as_uint(tn_) << 32);   \
        ++sp;                                                                                             \
    } while (0)
#define RT_STACK_POP(out_)                                                                                \
    do {                                                                                                  \
        --sp;                                                                                             \
        if (sp < RT_LDS_STACK) (out_) = rt_lds_stack[sp * RT_BLOCK_DIM + threadIdx.x];                    \
        else { const unsigned long long w_ = spill[sp - RT_LDS_STACK]; (out_) = make_uint2((uint32_t)w_, (uint32_t)(w_ >> 32)); } \
    } while (0)

typedef const RT_GLOBAL DNode* GNodePtr;
typedef const RT_GLOBAL DTri* GTriPtr;

enum { TRAV_CLOSEST = 0, TRAV_TWO = 1, TRAV_ANY = 2 };

// The reference gathers every triangle of every leaf whose chain of boxes the
// ray hits, intersects them all and stable-sorts by t (mesh.rs:326-376).  The
// walk below visits the nearer child first and skips boxes entered beyond the
// current answer; it returns the same (t, triangle) because a triangle inside a
// box cannot be hit before the box is entered, and ties keep the lower
// leaf-order index, which is the stable sort's order.
//   TRAV_CLOSEST: nearest hit with t < bound (strict).
//   TRAV_TWO:     nearest and second nearest (Hit::far, mesh.rs:411-423).
//   TRAV_ANY:     true as soon as one hit has t <= bound.
#define RT_POP 0xFFFFFFFFu   /* "nothing left": has the leaf bit set, so it also ends the inner-node loop */

// compare-exchange of (distance, reference) pairs: afterwards ta <= tb
#define RT_CSWAP(ta, ra, tb, rb)                                              \
    do {                                                                      \
        const bool sw_ = (tb) < (ta);                                         \
        const float tx_ = sw_ ? (tb) : (ta), ty_ = sw_ ? (ta) : (tb);         \
        const uint32_t rx_ = sw_ ? (rb) : (ra), ry_ = sw_ ? (ra) : (rb);      \
        (ta) = tx_; (tb) = ty_; (ra) = rx_; (rb) = ry_;                       \
    } while (0)

// A walk can be restricted (SEG): it then enters only boxes whose slab interval [tn, tf] meets [lo, hi], and gives
// up after `budget` inner nodes (0 = never), setting `aborted`.  Restricting the BOXES does not restrict the hits: a
// triangle found in a visited leaf counts wherever along the ray it is hit.  So K walks over K intervals that
// cover the ray together test every triangle the unrestricted walk tests, and the best of their answers (nearest,
// ties to the lower leaf-order index; or "any") is the unrestricted answer -- a long ray can be cut into pieces
// that run in parallel lanes.
struct Seg { float lo, hi; uint32_t budget; };

template <int MODE, bool SEG>
__device__ bool mesh_traverse_seg(const DMesh& M, const Ray& ray, double bound, MeshHit& mh, Visits& vc, const Seg seg, bool& aborted) {
    mh.n = 0; mh.t0 = bound; mh.t1 = bound; mh.r0 = 0;   // r0 = 0: a hit AT an outside bound never wins the tie
    float tn_root;
    uint32_t steps = 0;
    if (!node_box(M.root_min, M.root_max, ray, tn_root)) return false;

    int sp = 0;
    volatile unsigned long long spill[RT_MAX_STACK - RT_LDS_STACK];
    const GNodePtr nodes = gptr(M.nodes);
    const GTriPtr tris = gptr(M.tris);
    uint32_t cur = M.root_ref;
    // boxes entered beyond `limit` cannot hold the answer; kept in f32, rounded up so that the skip stays safe
    float limit = limit_up(bound);
    const float INF = __uint_as_float(0x7F800000u);
    const PRay pr = pray_new_d(ray.o, ray.d, M.coord_max);   // the boxes as one packed FMA per two planes, slack in the constants

    while (cur != RT_POP) {
        // ---- inner nodes, until this lane holds a leaf (the lanes of a wave leave this loop together, so
        //      the triangle tests below run with as many lanes as possible)
        while (!(cur & RT_LEAF_BIT)) {
            const GNodePtr nd = &nodes[cur];
            vc.nodes++; vc.units += 7;
            if (SEG && seg.budget && ++steps > seg.budget) { aborted = true; return false; }
            float t0, t1, t2, t3, f0, f1, f2, f3;
            RT_CHILD_PAIR_F((*nd), 0, t0, f0, t1, f1);
            RT_CHILD_PAIR_F((*nd), 2, t2, f2, t3, f3);
            bool h0 = !(t0 > f0 || f0 < 0.f), h1 = !(t1 > f1 || f1 < 0.f), h2 = !(t2 > f2 || f2 < 0.f), h3 = !(t3 > f3 || f3 < 0.f);
            if (SEG) {
                h0 = h0 && !(t0 > seg.hi) && !(f0 < seg.lo); h1 = h1 && !(t1 > seg.hi) && !(f1 < seg.lo);
                h2 = h2 && !(t2 > seg.hi) && !(f2 < seg.lo); h3 = h3 && !(t3 > seg.hi) && !(f3 < seg.lo);
            }
            uint32_t r0 = nd->ref[0], r1 = nd->ref[1], r2 = nd->ref[2], r3 = nd->ref[3];
            if (r2 == RT_NO_CHILD) h2 = false;
            if (r3 == RT_NO_CHILD) h3 = false;
            // a child that is missed, entered beyond the limit, or absent sorts last with distance +inf
            t0 = (h0 && !(t0 > limit)) ? t0 : INF;
            t1 = (h1 && !(t1 > limit)) ? t1 : INF;
            t2 = (h2 && !(t2 > limit)) ? t2 : INF;
            t3 = (h3 && !(t3 > limit)) ? t3 : INF;
            if (MODE != TRAV_ANY) {
                // nearest first (a shadow walk does not care about the order)
                RT_CSWAP(t0, r0, t1, r1); RT_CSWAP(t2, r2, t3, r3);
                RT_CSWAP(t0, r0, t2, r2); RT_CSWAP(t1, r1, t3, r3);
                RT_CSWAP(t1, r1, t2, r2);
                if (t3 < INF) RT_STACK_PUSH(r3, t3);
                if (t2 < INF) RT_STACK_PUSH(r2, t2);
                if (t1 < INF) RT_STACK_PUSH(r1, t1);
                cur = t0 < INF ? r0 : RT_POP;
            } else {
                cur = RT_POP;
                float tc = INF;
                if (t3 < INF) { cur = r3; tc = t3; }
                if (t2 < INF) { if (cur != RT_POP) RT_STACK_PUSH(cur, tc); cur = r2; tc = t2; }
                if (t1 < INF) { if (cur != RT_POP) RT_STACK_PUSH(cur, tc); cur = r1; tc = t1; }
                if (t0 < INF) { if (cur != RT_POP) RT_STACK_PUSH(cur, tc); cur = r0; tc = t0; }
            }
            if (cur == RT_POP) {
                while (sp > 0) {
                    uint2 e;
                    RT_STACK_POP(e);
                    if (!(__uint_as_float(e.y) > limit)) { cur = e.x; break; }
                }
            }
        }
        if (cur == RT_POP) break;
        // ---- one leaf: its triangles, exact f64 (mesh.rs:25-59)
        uint32_t rank = cur & ~RT_LEAF_BIT;
        while (true) {
            const TriData T = tri_load(&tris[rank]);
            const uint32_t last = T.last;
            double t; float u, v;
            vc.tris++; vc.units += 5;
            if (tri_test(T, ray, t, u, v)) {
                if (MODE == TRAV_ANY) {
                    if (t <= bound) return true;
                } else if (MODE == TRAV_CLOSEST) {
                    if (t < mh.t0 || (t == mh.t0 && rank < mh.r0)) {
                        mh.t0 = t; mh.r0 = rank; mh.u = u; mh.v = v; mh.n = 1;
                        limit = limit_up(t);
                    }
                } else {
                    if (mh.n == 0 || t < mh.t0 || (t == mh.t0 && rank < mh.r0)) {
                        if (mh.n >= 1) { mh.t1 = mh.t0; mh.n = 2; } else mh.n = 1;
                        mh.t0 = t; mh.r0 = rank; mh.u = u; mh.v = v;
                    } else if (mh.n == 1 || t < mh.t1) { mh.t1 = t; mh.n = 2; }
                    if (mh.n == 2) limit = limit_up(mh.t1);
                }
            }
            if (last) break;
            ++rank;
        }
        cur = RT_POP;
        while (sp > 0) {
            uint2 e;
            RT_STACK_POP(e);
            if (!(__uint_as_float(e.y) > limit)) { cur = e.x; break; }
        }
    }
    return mh.n > 0;
}
template <int MODE>
__device__ bool mesh_traverse(const DMesh& M, const Ray& ray, double bound, MeshHit& mh, Visits& vc) {
    Seg none; none.lo = 0.f; none.hi = 0.f; none.budget = 0;
    bool aborted = false;
    return mesh_traverse_seg<MODE, false>(M, ray, bound, mh, vc, none, aborted);
}

// The piece of a ray's passage through a mesh's root box that lane `k` of `K` walks (the first piece is open
// towards -inf and the last towards +inf, so that together they cover every box the plain walk can enter)
DEV Seg mesh_segment(const DMesh& M, const Ray& ray, double bound, uint32_t k, uint32_t K) {
    const float INF = __uint_as_float(0x7F800000u);
    float tn, tf;
    Seg sg; sg.lo = -INF; sg.hi = INF; sg.budget = 0;
    if (!node_box2(M.root_min, M.root_max, ray, tn, tf)) return sg;   // the walk returns at its own root test
    const float a = fmaxf(tn, 0.f), b = fminf(tf, limit_up(bound));
    if (!(b > a)) return sg;
    const float step = (b - a) / (float)K;
    if (k > 0) sg.lo = a + step * (float)k;
    if (k + 1 < K) sg.hi = a + step * (float)(k + 1);
    return sg;
}

// mesh.rs:348-439 from a traversal result (nearest triangle, its barycentrics,
// second-nearest t)
DEV void mesh_hit_fill(const DMesh& M, const Ray& ray, uint32_t rank, double t0, double t1, float bu, float bv, Hit& h) {
    const uint32_t tri = gptr(M.tris)[rank].orig;
    const RT_GLOBAL uint32_t* n = gptr(M.tri_nrm) + 3 * (size_t)tri;
    const RT_GLOBAL double* nrm = gptr(M.normals);
    {   // triangle_intersect_normal, mesh.rs:79-89 (f32 weights, widened)
        float u = 1.f - bu - bv, v = bu, w = bv;
        h.normal = ld3(nrm + 3 * (size_t)n[0]) * (double)u + ld3(nrm + 3 * (size_t)n[1]) * (double)v + ld3(nrm + 3 * (size_t)n[2]) * (double)w;
    }
    h.near = t0; h.vnear = along(ray, t0);
    h.far = t1; h.vfar = along(ray, t1);
    if (M.tri_uv != nullptr && M.n_texcoords != 0) {   // triangle_intersect_uvs, mesh.rs:61-77
        const RT_GLOBAL uint32_t* c = gptr(M.tri_uv) + 3 * (size_t)tri;
        const RT_GLOBAL float* a = gptr(M.texcoords) + 2 * (size_t)c[0];
        const RT_GLOBAL float* b = gptr(M.texcoords) + 2 * (size_t)c[1];
        const RT_GLOBAL float* cc = gptr(M.texcoords) + 2 * (size_t)c[2];
        float iu = bu, iv = bv, iw = 1.f - bu - bv;
        float uu = a[0] * iw + b[0] * iu + cc[0] * iv;
        float vv = a[1] * iw + b[1] * iu + cc[1] * iv;
        h.u = rem_euclid_f(uu, 1.f);
        h.v = rem_euclid_f(1.f - vv, 1.f);
    } else { h.u = 0.f; h.v = 0.f; }
    h.prim = tri;
}

// Mesh::intersect as a whole (used for the refraction helper rays, scene.rs:143-147)
__device__ bool mesh_intersect(const DMesh& M, const Ray& ray, Hit& h, Visits& vc) {
    MeshHit mh;
    if (!mesh_traverse<TRAV_TWO>(M, ray, INFINITY, mh, vc)) return false;
    mesh_hit_fill(M, ray, mh.r0, mh.t0, mh.n >= 2 ? mh.t1 : mh.t0, mh.u, mh.v, h);
    return true;
}

// ---------------------------------------------------------------- scene casts
// object.intersect(ray) for object oi (object/mod.rs:61-64)
// MESHES = false: the caller knows the object is no mesh, and the mesh walk (registers, an LDS stack) stays out of
// its kernel.
template <bool MESHES = true>
__device__ bool object_intersect(const DScene& S, uint32_t oi, const Ray& ray, Hit& h, Visits& vc) {
    const RT_GLOBAL DObject* op = gptr(S.objects) + oi;
    const uint32_t kind = op->kind, payload = op->payload;
    switch (kind) {
        case 0: return sphere_intersect(gptr(S.spheres) + 4 * (size_t)payload, ray, h);
        case 1: return plane_intersect(gptr(S.planes) + 8 * (size_t)payload, ray, h);
        case 2: return box_intersect(ld3(gptr(S.boxes) + 6 * (size_t)payload), ld3(gptr(S.boxes) + 6 * (size_t)payload + 3), ray, h);
        default: return MESHES ? mesh_intersect(S.meshes[payload], ray, h, vc) : false;
    }
}

struct Closest { uint32_t obj, rank; double t, t2; float u, v; };

// ---- object tree (DTlasNode).  The boxes cull exactly as long as the ray direction is a unit vector: the
// reference's sphere test (sphere.rs:33-48) is only the geometric one for unit directions, and a ray reflected off
// unnormalised mesh normals is not unit (mesh.rs:79-89).  For such a ray every box counts as hit and the walk
// degenerates into the reference's linear scan.
DEV bool tlas_culls(const Ray& ray) { return fabs(dot(ray.d, ray.d) - 1.) < 1e-9; }
DEV bool tlas_object_near(const DScene& S, uint32_t oi, const Ray& ray, double& near) {
    const RT_GLOBAL DObject* op = gptr(S.objects) + oi;
    const uint32_t kind = op->kind, payload = op->payload;
    if (kind == 0) return sphere_near(gptr(S.spheres) + 4 * (size_t)payload, ray, near);
    return box_near(gptr(S.boxes) + 6 * (size_t)payload, ray, near);
}
// the object behind a leaf reference of a DTlasHalf record (its numbers by object index, DScene::tlas_geom)
DEV bool tlas_leaf_near(const DScene& S, uint32_t ref, const Ray& ray, double& near) {
    const RT_GLOBAL double* g = gptr(S.tlas_geom) + 6 * (size_t)(ref & ~(RT_TLAS_LEAF | RT_TLAS_BOX));
    if (!(ref & RT_TLAS_BOX)) return sphere_near(g, ray, near);
    return box_near(g, ray, near);
}
// cast_ray_once over the tree's objects: nearest by `near`, ties to the lower object index (scene.rs:57-82); `c` may
// already hold the best of the objects outside the tree
__device__ bool tlas_closest(const DScene& S, const Ray& ray, Closest& c, bool any) {
    const RT_GLOBAL DTlasNode* nodes = (const RT_GLOBAL DTlasNode*)S.tlas;
    const bool cull = tlas_culls(ray);
    const uint32_t n = S.n_tlas;
    for (uint32_t i = 0; i < n;) {
        const RT_GLOBAL DTlasNode* nd = nodes + i;
        float tn;
        if (cull && !(node_box(nd->bmin, nd->bmax, ray, tn) && !(any && tn > limit_up(c.t)))) { i = nd->skip; continue; }
        const uint32_t oi = nd->object;
        if (!(oi & RT_TLAS_INNER)) {
            double near;
            if (tlas_object_near(S, oi, ray, near) && (!any || near < c.t || (near == c.t && oi < c.obj))) {
                c.obj = oi; c.rank = RT_NONE; c.t = near; c.t2 = 0.; c.u = 0.f; c.v = 0.f; any = true;
            }
        }
        i++;
    }
    return any;
}
__device__ bool tlas_any(const DScene& S, const Ray& ray, bool limited, double dist) {
    const RT_GLOBAL DTlasNode* nodes = (const RT_GLOBAL DTlasNode*)S.tlas;
    const bool cull = tlas_culls(ray);
    const uint32_t n = S.n_tlas;
    for (uint32_t i = 0; i < n;) {
        const RT_GLOBAL DTlasNode* nd = nodes + i;
        float tn;
        if (cull && !node_box(nd->bmin, nd->bmax, ray, tn)) { i = nd->skip; continue; }
        const uint32_t oi = nd->object;
        if (!(oi & RT_TLAS_INNER)) {
            double near;
            if (tlas_object_near(S, oi, ray, near) && (!limited || near <= dist)) return true;
        }
        i++;
    }
    return false;
}
// The four children of a four-wide record: each one's entry distance if the ray enters its box no later than `limit`,
// +inf if not (or if the slot is unused); without culling every child counts as entered at -inf.  (pr: the lane's
// PRay for the tree; one packed FMA per two planes, the rounding slack folded into its constants.)
#define RT_TQ_CHILDREN(t0_, t1_, t2_, t3_)                                                                                 \
    do {                                                                                                                  \
        if (!cull) { (t0_) = (t1_) = (t2_) = (t3_) = -INF; }                                                              \
        else {                                                                                                            \
            float f0_, f1_, f2_, f3_;                                                                                     \
            RT_CHILD_PAIR_F(W, 0, t0_, f0_, t1_, f1_);                                                                    \
            RT_CHILD_PAIR_F(W, 2, t2_, f2_, t3_, f3_);                                                                    \
            (t0_) = (!((t0_) > f0_ || f0_ < 0.f) && !((t0_) > limit)) ? (t0_) : INF;                                      \
            (t1_) = (!((t1_) > f1_ || f1_ < 0.f) && !((t1_) > limit)) ? (t1_) : INF;                                      \
            (t2_) = (!((t2_) > f2_ || f2_ < 0.f) && !((t2_) > limit)) ? (t2_) : INF;                                      \
            (t3_) = (!((t3_) > f3_ || f3_ < 0.f) && !((t3_) > limit)) ? (t3_) : INF;                                      \
        }                                                                                                                 \
        if (W.ref[0] == RT_TLAS_NONE) (t0_) = INF;                                                                        \
        if (W.ref[1] == RT_TLAS_NONE) (t1_) = INF;                                                                        \
        if (W.ref[2] == RT_TLAS_NONE) (t2_) = INF;                                                                        \
        if (W.ref[3] == RT_TLAS_NONE) (t3_) = INF;                                                                        \
    } while (0)
// compare-exchange of (distance, reference) pairs: afterwards ta <= tb
#define RT_CSWAP2(ta, ra, tb, rb)                                             \
    do {                                                                      \
        const bool sw_ = (tb) < (ta);                                         \
        const float tx_ = sw_ ? (tb) : (ta), ty_ = sw_ ? (ta) : (tb);         \
        const uint32_t rx_ = sw_ ? (rb) : (ra), ry_ = sw_ ? (ra) : (rb);      \
        (ta) = tx_; (tb) = ty_; (ra) = rx_; (rb) = ry_;                       \
    } while (0)
// two half-precision floats of one word
DEV void half_pair(uint32_t w, float& a, float& b) {
    a = __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu)));
    b = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
}
// one 64-byte record (DTlasHalf) by four 16-byte loads, its boxes widened to f32 (exactly: every half is a float)
typedef unsigned int __attribute__((ext_vector_type(4))) rt_w4;
struct TlasRec { float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4]; uint32_t ref[4]; };
#define RT_TQ_LOAD(cur_)                                                                      \
    const RT_GLOBAL rt_w4* p_ = (const RT_GLOBAL rt_w4*)(halves + (cur_));                    \
    const rt_w4 q0_ = p_[0], q1_ = p_[1], q2_ = p_[2], q3_ = p_[3];                           \
    TlasRec W;                                                                                \
    half_pair(q0_.x, W.minx[0], W.minx[1]); half_pair(q0_.y, W.minx[2], W.minx[3]);           \
    half_pair(q0_.z, W.miny[0], W.miny[1]); half_pair(q0_.w, W.miny[2], W.miny[3]);           \
    half_pair(q1_.x, W.minz[0], W.minz[1]); half_pair(q1_.y, W.minz[2], W.minz[3]);           \
    half_pair(q1_.z, W.maxx[0], W.maxx[1]); half_pair(q1_.w, W.maxx[2], W.maxx[3]);           \
    half_pair(q2_.x, W.maxy[0], W.maxy[1]); half_pair(q2_.y, W.maxy[2], W.maxy[3]);           \
    half_pair(q2_.z, W.maxz[0], W.maxz[1]); half_pair(q2_.w, W.maxz[2], W.maxz[3]);           \
    W.ref[0] = q3_.x; W.ref[1] = q3_.y; W.ref[2] = q3_.z; W.ref[3] = q3_.w

// tlas_closest for the kernels that own a per-lane stack (the lane-by-lane trace kernels): the four-wide records,
// nearer children first, so that the nearest object is found early and the boxes behind it are never entered.  Same
// answer as the skip-link walk: every object whose box the ray enters before the best distance is tested, ties go to
// the lower index.
__device__ bool tlas_closest_ordered(const DScene& S, const Ray& ray, Closest& c, bool any) {
    const RT_GLOBAL DTlasNode* nodes = (const RT_GLOBAL DTlasNode*)S.tlas;
    const RT_GLOBAL DTlasHalf* halves = (const RT_GLOBAL DTlasHalf*)S.tlas_half;
    const bool cull = tlas_culls(ray);
    const float INF = __uint_as_float(0x7F800000u);
    int sp = 0;
    volatile unsigned long long spill[RT_MAX_STACK - RT_LDS_STACK];
    float limit = any ? limit_up(c.t) : INF;
    float tn_root = -INF;
    if (cull && !(node_box(nodes[0].bmin, nodes[0].bmax, ray, tn_root) && !(tn_root > limit))) return any;
    const PRay pr = pray_new_d(ray.o, ray.d, S.tlas_half_coord_max);
    // an object reached through a leaf reference: nearest by `near`, ties to the lower index
    auto object = [&](uint32_t ref) {
        const uint32_t oi = ref & ~(RT_TLAS_LEAF | RT_TLAS_BOX);
        double near;
        if (tlas_leaf_near(S, ref, ray, near) && (!any || near < c.t || (near == c.t && oi < c.obj))) {
            c.obj = oi; c.rank = RT_NONE; c.t = near; c.t2 = 0.; c.u = 0.f; c.v = 0.f; any = true;
            limit = limit_up(near);
        }
    };
    uint32_t cur = 0;
    while (true) {
        RT_TQ_LOAD(cur);
        float t0, t1, t2, t3;
        RT_TQ_CHILDREN(t0, t1, t2, t3);
        uint32_t r0 = W.ref[0], r1 = W.ref[1], r2 = W.ref[2], r3 = W.ref[3];
        RT_CSWAP2(t0, r0, t1, r1); RT_CSWAP2(t2, r2, t3, r3);
        RT_CSWAP2(t0, r0, t2, r2); RT_CSWAP2(t1, r1, t3, r3);
        RT_CSWAP2(t1, r1, t2, r2);
        // leaves are tested on the spot, nearest first; what they find may close the children behind them
#define RT_TQ_LEAF(t_, r_) if ((t_) < INF && ((r_) & RT_TLAS_LEAF)) { if (!((t_) > limit)) object(r_); (t_) = INF; }
        RT_TQ_LEAF(t0, r0) RT_TQ_LEAF(t1, r1) RT_TQ_LEAF(t2, r2) RT_TQ_LEAF(t3, r3)
#undef RT_TQ_LEAF
        // the inner children still open, farthest first: the nearest is walked next, the others wait on the stack
        if (sp + 3 > RT_MAX_STACK) return any;   // cannot happen: upload refuses a tree this deep
        cur = RT_POP;
        float cur_t = INF;
#define RT_TQ_INNER(t_, r_) if ((t_) < INF && !((t_) > limit)) { if (cur != RT_POP) RT_STACK_PUSH(cur, cur_t); cur = (r_); cur_t = (t_); }
        RT_TQ_INNER(t3, r3) RT_TQ_INNER(t2, r2) RT_TQ_INNER(t1, r1) RT_TQ_INNER(t0, r0)
#undef RT_TQ_INNER
        if (cur == RT_POP) {
            while (sp > 0) {
                uint2 e;
                RT_STACK_POP(e);
                if (!(__uint_as_float(e.y) > limit)) { cur = e.x; break; }
            }
            if (cur == RT_POP) break;
        }
    }
    return any;
}

// Scene::cast_ray_once (scene.rs:57-82): every object in scene order; the
// stable sort by `near` keeps the lowest index among equals, i.e. a later
// object must be strictly nearer to win.
// FAR = false: the caller knows that no mesh of the scene needs Hit::far, and the two-hit walk is compiled out.
template <bool FAR = true>
__device__ bool cast_closest(const DScene& S, const Ray& ray, Closest& c, Visits& vc) {
    c.obj = RT_NONE; c.rank = RT_NONE; c.t = 0.; c.t2 = 0.; c.u = 0.f; c.v = 0.f;
    bool any = false;
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        struct { uint32_t kind, payload, need_far; } o;
        o.kind = objs[oi].kind; o.payload = objs[oi].payload; o.need_far = objs[oi].need_far;
        double near; bool hit;
        if (o.kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)o.payload, ray, near);
        else if (o.kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)o.payload, ray, near);
        else if (o.kind == 2) hit = box_near(gptr(S.boxes) + 6 * (size_t)o.payload, ray, near);
        else {
            MeshHit mh;
            if (FAR && o.need_far) {
                hit = mesh_traverse<TRAV_TWO>(S.meshes[o.payload], ray, INFINITY, mh, vc);
                if (hit && (!any || mh.t0 < c.t)) {
                    c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.n >= 2 ? mh.t1 : mh.t0; c.u = mh.u; c.v = mh.v; any = true;
                }
            } else {
                hit = mesh_traverse<TRAV_CLOSEST>(S.meshes[o.payload], ray, any ? c.t : INFINITY, mh, vc);
                if (hit) { c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.t0; c.u = mh.u; c.v = mh.v; any = true; }
            }
            continue;
        }
        if (hit && (!any || near < c.t)) { c.obj = oi; c.rank = RT_NONE; c.t = near; any = true; }
    }
    if (S.n_tlas) any = tlas_closest_ordered(S, ray, c, any);
    return any;
}

// cast_closest in a scene without meshes
__device__ bool cast_closest_analytic(const DScene& S, const Ray& ray, Closest& c) {
    c.obj = RT_NONE; c.rank = RT_NONE; c.t = 0.; c.t2 = 0.; c.u = 0.f; c.v = 0.f;
    bool any = false;
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        const uint32_t kind = objs[oi].kind, payload = objs[oi].payload;
        double near; bool hit;
        if (kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)payload, ray, near);
        else if (kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)payload, ray, near);
        else if (kind == 2) hit = box_near(gptr(S.boxes) + 6 * (size_t)payload, ray, near);
        else continue;
        if (hit && (!any || near < c.t)) { c.obj = oi; c.rank = RT_NONE; c.t = near; any = true; }
    }
    if (S.n_tlas) any = tlas_closest_ordered(S, ray, c, any);
    return any;
}

// cast_closest with every plain mesh walk restricted to `sg_of(mesh)`: `k`/`K` pick the piece (K = 1: the whole
// ray), `budget` bounds each walk.  aborted = some walk gave up; the result is then incomplete.
__device__ bool cast_closest_seg(const DScene& S, const Ray& ray, Closest& c, Visits& vc, uint32_t k, uint32_t K, uint32_t budget, bool& aborted) {
    c.obj = RT_NONE; c.rank = RT_NONE; c.t = 0.; c.t2 = 0.; c.u = 0.f; c.v = 0.f;
    bool any = false;
    aborted = false;
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        struct { uint32_t kind, payload, need_far; } o;
        o.kind = objs[oi].kind; o.payload = objs[oi].payload; o.need_far = objs[oi].need_far;
        double near; bool hit;
        if (o.kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)o.payload, ray, near);
        else if (o.kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)o.payload, ray, near);
        else if (o.kind == 2) hit = box_near(gptr(S.boxes) + 6 * (size_t)o.payload, ray, near);
        else {
            MeshHit mh;
            if (o.need_far) {
                hit = mesh_traverse<TRAV_TWO>(S.meshes[o.payload], ray, INFINITY, mh, vc);
                if (hit && (!any || mh.t0 < c.t)) {
                    c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.n >= 2 ? mh.t1 : mh.t0; c.u = mh.u; c.v = mh.v; any = true;
                }
            } else {
                double bound = any ? c.t : INFINITY;
                // The K lanes of a group cut ONE interval into pieces, so they must agree on it: the best distance so
                // far is the group's, not the lane's (each lane only knows what its own pieces of the earlier objects
                // held; pieces of different intervals would leave gaps).  A hit at exactly that distance in this, later,
                // object loses either way.
                if (K > 1) for (uint32_t off = 1; off < K; off <<= 1) bound = fmin(bound, __shfl_xor(bound, (int)off));
                Seg sg;
                if (K > 1) sg = mesh_segment(S.meshes[o.payload], ray, bound, k, K);
                else { sg.lo = -INFINITY; sg.hi = INFINITY; }
                sg.budget = budget;
                bool ab = false;
                hit = mesh_traverse_seg<TRAV_CLOSEST, true>(S.meshes[o.payload], ray, bound, mh, vc, sg, ab);
                if (ab) { aborted = true; return false; }
                if (hit) { c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.t0; c.u = mh.u; c.v = mh.v; any = true; }
            }
            continue;
        }
        if (hit && (!any || near < c.t)) { c.obj = oi; c.rank = RT_NONE; c.t = near; any = true; }
    }
    if (S.n_tlas) any = tlas_closest_ordered(S, ray, c, any);
    return any;
}

// cast_any with every mesh walk restricted to piece k of K
__device__ bool cast_any_seg(const DScene& S, const Ray& ray, bool limited, double dist, Visits& vc, uint32_t k, uint32_t K) {
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        struct { uint32_t kind, payload; } o;
        o.kind = objs[oi].kind; o.payload = objs[oi].payload;
        double near; bool hit;
        if (o.kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)o.payload, ray, near);
        else if (o.kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)o.payload, ray, near);
        else if (o.kind == 2) hit = box_near(gptr(S.boxes) + 6 * (size_t)o.payload, ray, near);
        else {
            MeshHit mh;
            const double bound = limited ? dist : INFINITY;
            Seg sg = mesh_segment(S.meshes[o.payload], ray, bound, k, K);
            bool ab = false;
            if (mesh_traverse_seg<TRAV_ANY, true>(S.meshes[o.payload], ray, bound, mh, vc, sg, ab)) return true;
            continue;
        }
        if (hit && (!limited || near <= dist)) return true;
    }
    return S.n_tlas != 0 && tlas_any(S, ray, limited, dist);
}

// tlas_any for a kernel that owns a per-lane stack: the four-wide records, nothing beyond `dist` entered.
__device__ bool tlas_any_ordered(const DScene& S, const Ray& ray, bool limited, double dist) {
    const RT_GLOBAL DTlasNode* nodes = (const RT_GLOBAL DTlasNode*)S.tlas;
    const RT_GLOBAL DTlasHalf* halves = (const RT_GLOBAL DTlasHalf*)S.tlas_half;
    const bool cull = tlas_culls(ray);
    const float INF = __uint_as_float(0x7F800000u);
    int sp = 0;
    volatile unsigned long long spill[RT_MAX_STACK - RT_LDS_STACK];
    const float limit = limited ? limit_up(dist) : INF;
    float tn_root = -INF;
    if (cull && !(node_box(nodes[0].bmin, nodes[0].bmax, ray, tn_root) && !(tn_root > limit))) return false;
    const PRay pr = pray_new_d(ray.o, ray.d, S.tlas_half_coord_max);
    uint32_t cur = 0;
    while (true) {
        RT_TQ_LOAD(cur);
        float t0, t1, t2, t3;
        RT_TQ_CHILDREN(t0, t1, t2, t3);
        double near;
#define RT_TQ_LEAF(k, t_)                                                                                   \
        if ((t_) < INF && (W.ref[k] & RT_TLAS_LEAF)) {                                                      \
            if (tlas_leaf_near(S, W.ref[k], ray, near) && (!limited || near <= dist)) return true;          \
            (t_) = INF;                                                                                     \
        }
        RT_TQ_LEAF(0, t0) RT_TQ_LEAF(1, t1) RT_TQ_LEAF(2, t2) RT_TQ_LEAF(3, t3)
#undef RT_TQ_LEAF
        // the inner children entered, in the record's order (sorting them by distance costs more than it saves here)
        if (sp + 3 > RT_MAX_STACK) return false;   // cannot happen: upload refuses a tree this deep
        cur = RT_POP;
#define RT_TQ_INNER(k, t_) if ((t_) < INF) { if (cur != RT_POP) RT_STACK_PUSH(cur, 0.f); cur = W.ref[k]; }
        RT_TQ_INNER(3, t3) RT_TQ_INNER(2, t2) RT_TQ_INNER(1, t1) RT_TQ_INNER(0, t0)
#undef RT_TQ_INNER
        if (cur == RT_POP) {
            if (sp == 0) break;
            uint2 e;
            RT_STACK_POP(e);
            cur = e.x;
        }
    }
    return false;
}
#undef RT_TQ_CHILDREN
#undef RT_TQ_LOAD

// Shadow cast of one lane in a scene without meshes (sun.rs:89: any object the ray hits at all; point.rs:91-92: the
// nearest hit has near <= dist, i.e. some object has near <= dist).
__device__ bool cast_any_analytic(const DScene& S, const Ray& ray, bool limited, double dist) {
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        const uint32_t kind = objs[oi].kind, payload = objs[oi].payload;
        double near; bool hit;
        if (kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)payload, ray, near);
        else if (kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)payload, ray, near);
        else if (kind == 2) hit = box_near(gptr(S.boxes) + 6 * (size_t)payload, ray, near);
        else continue;
        if (hit && (!limited || near <= dist)) return true;
    }
    return S.n_tlas != 0 && tlas_any_ordered(S, ray, limited, dist);
}

// Rebuilds the full Hit of the closest object from the trace record
// BOXES = false: the caller vouches that the scene has no cube (and `ray` may come without 1/direction)
template <bool MESHES = true, bool BOXES = true>
__device__ void closest_hit_fill(const DScene& S, const Ray& ray, uint32_t obj, uint32_t rank, double t, double t2, float u, float v, Hit& h) {
    const RT_GLOBAL DObject* op = gptr(S.objects) + obj;
    const uint32_t kind = op->kind, payload = op->payload;
    if (MESHES && kind == 3) mesh_hit_fill(S.meshes[payload], ray, rank, t, t2, u, v, h);
    else if (kind == 0) sphere_intersect(gptr(S.spheres) + 4 * (size_t)payload, ray, h);
    else if (kind == 1) plane_intersect(gptr(S.planes) + 8 * (size_t)payload, ray, h);
    else if (BOXES) box_intersect(ld3(gptr(S.boxes) + 6 * (size_t)payload), ld3(gptr(S.boxes) + 6 * (size_t)payload + 3), ray, h);
}

// ---------------------------------------------------------------- material.rs:119-141
template <typename P> DEV uint32_t ldrgb(P p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
DEV uint32_t texture_at(const DMaterial& m, float u, float v) {
    if (m.tex_kind == 0) return m.a_rgb;
    if (m.tex_kind == 1) {
        float w = (float)m.tex_w, h = (float)m.tex_h;
        uint32_t x = as_u32f(clampf(u * w, 0.f, w - 1.f));
        uint32_t y = as_u32f(clampf(v * h, 0.f, h - 1.f));
        return ldrgb(gptr(m.tex_rgb) + 3 * ((size_t)y * m.tex_w + x));
    }
    bool a = u > 0.5f, b = v > 0.5f;
    return a == b ? m.a_rgb : m.b_rgb;
}

// ---------------------------------------------------------------- skybox.rs
DEV uint32_t sky_px(const DScene& S, double xx, double yy) {
    uint32_t xi = as_u32d(xx), yi = as_u32d(yy);
    if (xi >= S.sky_w || yi >= S.sky_h) return 0u;   // the reference's get_pixel would panic
    return ldrgb(gptr(S.sky_pix) + 3 * ((size_t)yi * S.sky_w + xi));
}
__device__ uint32_t skybox_color(const DScene& S, D3 dir) {
    if (S.sky_kind == 0) return S.sky_rgb;                                                       // skybox.rs:17-21
    if (S.sky_kind == 1) return newf(dir.x / 2. + 0.5, dir.y / 2. + 0.5, dir.z / 2. + 0.5);      // :27-31, material.rs:47-49
    // skybox.rs:89-119
    Ray r = ray_new(dir * 2., -dir);
    Hit h;
    if (!box_intersect(d3(0., 0., 0.), d3(0.5, 0.5, 0.5), r, h)) return 0u;
    D3 normal = h.normal;
    uint32_t cx, cy;
    if (normal.z == -1.) { cx = 1; cy = 1; }
    else if (normal.x == -1.) { cx = 2; cy = 1; }
    else if (normal.x == 1.) { cx = 0; cy = 1; }
    else if (normal.z == 1.) { cx = 3; cy = 1; }
    else if (normal.y == 1.) { cx = 1; cy = 0; }
    else if (normal.y == -1.) { cx = 1; cy = 2; }
    else return 0u;
    float u = h.u, v = h.v;
    if (normal.y == 1.) { u = 1.f - h.u; v = 1.f - h.v; }
    else if (normal.y == -1.) { u = h.u; v = 1.f - h.v; }
    // poll_tex, skybox.rs:64-86
    uint32_t cell = S.sky_w / 4;
    double x = (double)(cx * cell) + (double)u * (double)(cell - 1);
    double y = (double)(cy * cell) + (double)v * (double)(cell - 1);
    if (fmod(x, 1.) == 0. && fmod(y, 1.) == 0.) return sky_px(S, x, y);
    double xmin = floor(x), xmax = ceil(x), ymin = floor(y), ymax = ceil(y);
    D3 c00 = to_d3(sky_px(S, xmin, ymin)), c10 = to_d3(sky_px(S, xmax, ymin));
    D3 c01 = to_d3(sky_px(S, xmin, ymax)), c11 = to_d3(sky_px(S, xmax, ymax));
    double tx = (x - xmin) / (xmax - xmin), ty = (y - ymin) / (ymax - ymin);
    D3 a = c00 * (1. - tx) + c10 * tx;                                                           // blerp, math/mod.rs:22-26
    D3 b = c01 * (1. - tx) + c11 * tx;
    return to_color(a * (1. - ty) + b * ty);
}

// ---------------------------------------------------------------- camera.rs:55-67
DEV D3 camera_direction(const DCamera& S, double x, double y) {
    double nx = x - S.half_w;
    double ny = y - S.half_h;
    D3 f = normalize(d3(nx, -ny, -S.cam_z));
    // row 2 of from_forward(f) * rot (matrix.rs:38-57,148-185): the products are
    // accumulated from 0.0 in k order; forward() negates it (matrix.rs:113-115)
    double a0 = -f.x, a1 = -f.y, a2 = -f.z;
    double o0 = 0., o1 = 0., o2 = 0.;
    o0 += a0 * S.rot[0][0]; o0 += a1 * S.rot[1][0]; o0 += a2 * S.rot[2][0]; o0 += 0. * 0.;
    o1 += a0 * S.rot[0][1]; o1 += a1 * S.rot[1][1]; o1 += a2 * S.rot[2][1]; o1 += 0. * 0.;
    o2 += a0 * S.rot[0][2]; o2 += a1 * S.rot[1][2]; o2 += a2 * S.rot[2][2]; o2 += 0. * 0.;
    return d3(-o0, -o1, -o2);
}
