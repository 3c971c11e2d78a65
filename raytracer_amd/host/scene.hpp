// C++ mirror of the reference's Scene / Object / Light API for the render path
// (raytracer/src/{scene,camera,material,skybox}.rs, object/*.rs, lighting/*.rs).
// Names, fields, defaults and argument meaning follow the reference so that a
// user of `Scene::render` / `Scene::render_to` finds the same surface; the
// per-pixel loop behind render() runs on the MI355X through librtgpu.so.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtgpu.h"
#include "rmath.hpp"

namespace rth {

struct Color { uint8_t r = 0, g = 0, b = 0; Color() = default; Color(uint8_t r_, uint8_t g_, uint8_t b_) : r(r_), g(g_), b(b_) {} };  // material.rs:5-9

// material.rs:108-117
struct Texture {
    enum Kind { Solid = RT_TEX_SOLID, Image = RT_TEX_IMAGE, Checkerboard = RT_TEX_CHECKER } kind = Solid;
    Color a{255, 255, 255}, b{0, 0, 0};
    uint32_t w = 0, h = 0;
    std::shared_ptr<std::vector<uint8_t>> rgb;   // RGB8, row-major, top-left first
    static Texture solid(Color c) { Texture t; t.kind = Solid; t.a = c; return t; }
    static Texture checkerboard(Color a, Color b) { Texture t; t.kind = Checkerboard; t.a = a; t.b = b; return t; }
    static Texture image(uint32_t w, uint32_t h, const uint8_t* rgb);
};

// material.rs:147-176 (Material::default: white, 0, 0, ior 1.3, 0)
struct Material {
    Texture texture = Texture::solid(Color(255, 255, 255));
    double reflectiveness = 0., transparency = 0., ior = 1.3, emissivity = 0.;
};

// camera.rs:5-52
struct Camera {
    int32_t vw = 800, vh = 600;
    Vector3 origin;
    double yaw = 0., pitch = 0., fov = 60., chf = chf_of(60.);
    static double chf_of(double fov) { return std::tan((90. - fov * 0.5) * 0.017453); }   // camera.rs:45-47
    void set_fov(double f) { fov = f; chf = chf_of(f); }                                  // camera.rs:49-52
};

// scene.rs:17-32
struct SceneOptions { uint32_t max_ray_depth = 4; Color ambient{40, 40, 40}; };

// acceleration.rs:407-417, flattened (see rt_bvh_node)
struct FlatBvh {
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> prim_indices;
    uint32_t depth = 0;
};

// object/mesh.rs:91-119
struct Mesh {
    std::vector<Vector3> verts;
    std::vector<std::array<uint32_t, 3>> tris;
    std::vector<Vector3> normals;
    std::vector<std::array<uint32_t, 3>> tri_normals;
    std::vector<std::array<float, 2>> texcoords;
    std::vector<std::array<uint32_t, 3>> tri_texcoords;
    Material material;
    std::shared_ptr<FlatBvh> sbvh;
    bool device_bvh = false;               // leave the tree to the render library (rt_mesh::nodes == NULL: built on the device)

    static bool from_obj(const std::string& file_name, Material material, Mesh& out, std::string& err);   // mesh.rs:135-205
    void recalculate_normals();            // mesh.rs:208-254
    bool generate_sbvh(std::string& err);  // mesh.rs:257-272
    void shift(Vector3 delta);             // mesh.rs:275-277
    void center();                         // mesh.rs:279-293
    void scale(double delta);              // mesh.rs:296-298
    void rotate_xyz(Vector3 rot);          // mesh.rs:301-311
    void rotate_zyx(Vector3 rot);          // mesh.rs:314-324
};

struct Sphere { Vector3 origin; double radius = 0.; Material material; };                   // sphere.rs:12-21
struct Plane { Vector3 origin; Vector3 normal{0., 1., 0.}; Material material; float uv_wrap = 1.f; };  // plane.rs:11-47
struct Aabb { Vector3 pos, size; Material material; };                                      // aabb.rs:86-99 (size = half extent)

struct SceneObject {
    enum Kind { KSphere = RT_OBJ_SPHERE, KPlane = RT_OBJ_PLANE, KAabb = RT_OBJ_BOX, KMesh = RT_OBJ_MESH } kind;
    Sphere sphere; Plane plane; Aabb aabb; std::shared_ptr<Mesh> mesh;
    const Material& material() const {                                                       // object/mod.rs:67-70
        return kind == KSphere ? sphere.material : kind == KPlane ? plane.material : kind == KAabb ? aabb.material : mesh->material;
    }
};

// lighting/sun.rs:13-53, point.rs:12-47, area.rs:14-95
struct Light {
    enum Kind { Sun = RT_LIGHT_SUN, Point = RT_LIGHT_POINT, Area = RT_LIGHT_AREA } kind = Sun;
    Color color{255, 255, 255};
    double intensity = 1.;
    int32_t specular_power = 32;
    double specular_strength = 0.5;
    Vector3 vector{0., -1., 0.};      // Sun: direction; Point: position; Area sphere: centre
    bool shadows = true;
    double shadow_coefficient = 0.5;
    double max_distance = 50.;
    uint32_t area_surface = RT_AREA_SPHERE, area_iterations = 4;
    double area_radius = 0.;
    Vector3 area_corners[4];
    static Light sun() { return Light(); }
    static Light point() { Light l; l.kind = Point; l.intensity = 6.; l.specular_strength = 0.7; l.vector = Vector3(0., 0., 0.); return l; }
    static Light area() { Light l = point(); l.kind = Area; return l; }
};

// skybox.rs:15,25,35-61
struct Skybox {
    enum Kind { Solid = RT_SKY_SOLID, Normal = RT_SKY_NORMAL, Cubemap = RT_SKY_CUBEMAP } kind = Normal;
    Color color;
    uint32_t w = 0, h = 0;
    std::shared_ptr<std::vector<uint8_t>> rgb;
};

// scene.rs:35-53
class Scene {
   public:
    std::vector<SceneObject> objects;
    std::vector<Light> lights;
    Camera camera;
    Skybox skybox;
    SceneOptions options;

    ~Scene();
    // Flattens into the plain-data description of include/rtgpu.h.  The returned
    // pointer and everything it references stay valid until the next flatten()
    // or the Scene's destruction.
    const rt_scene_desc* flatten(std::string& err);

    // scene.rs:220-246: the whole frame as RGB8, row-major.  Runs on the GPU
    // through librtgpu.so; fails (false + err) when no device is available.
    bool render(std::vector<Color>& out, std::string& err, int device_id = 0);
    // scene.rs:249-265 (PNG only)
    bool render_to(const std::string& path, std::string& err, int device_id = 0);

    bool save(const std::string& path, std::string& err);

   private:
    struct Flat;
    Flat* flat_ = nullptr;
    rt_ctx* ctx_ = nullptr;
};

// Synthetic inputs (seeded, deterministic)
std::shared_ptr<Mesh> make_heightfield(uint32_t nx, uint32_t nz, double cell, double amplitude, double base_freq,
                                       uint32_t octaves, uint64_t seed);
void make_star_atlas(uint32_t cell, uint64_t seed, std::vector<uint8_t>& rgb);

bool write_png(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err);
bool read_png(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err);
bool read_jpeg(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err);   // baseline only
bool read_image(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err);  // PNG or JPEG, by content

}  // namespace rth
