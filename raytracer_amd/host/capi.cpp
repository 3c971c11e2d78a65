// Flat C API over the C++ scene classes, for the Python harness (ctypes).
// Every function returns 0 on success or -1 with the text in rth_last_error().
#include <cstring>

#include "scene.hpp"
#include "sdl.hpp"

namespace rth {
struct LoadedDesc;
bool save_desc(const rt_scene_desc* d, const std::string& path, std::string& err);
LoadedDesc* load_desc(const std::string& path, std::string& err);
void free_desc(LoadedDesc* l);
const rt_scene_desc* desc_of(LoadedDesc* l);
}  // namespace rth

using namespace rth;

static thread_local std::string g_err;
static int fail(const std::string& e) { g_err = e; return -1; }
static Vector3 V(const double* p) { return Vector3(p[0], p[1], p[2]); }
static Color C(const uint8_t* p) { return Color(p[0], p[1], p[2]); }

struct rth_material {   // mirrors Material (material.rs:147-164) + its texture
    uint32_t tex_kind; uint8_t a[3]; uint8_t b[3]; uint8_t _pad[2];
    uint32_t w, h; const uint8_t* rgb;
    double reflectiveness, transparency, ior, emissivity;
};
static Material M(const rth_material* m) {
    Material r;
    if (!m) return r;
    if (m->tex_kind == RT_TEX_IMAGE) r.texture = Texture::image(m->w, m->h, m->rgb);
    else if (m->tex_kind == RT_TEX_CHECKER) r.texture = Texture::checkerboard(C(m->a), C(m->b));
    else r.texture = Texture::solid(C(m->a));
    r.reflectiveness = m->reflectiveness; r.transparency = m->transparency; r.ior = m->ior; r.emissivity = m->emissivity;
    return r;
}

extern "C" {

const char* rth_last_error() { return g_err.c_str(); }

Scene* rth_scene_new() { return new Scene(); }
void rth_scene_free(Scene* s) { delete s; }

void rth_set_camera(Scene* s, int32_t vw, int32_t vh, const double* origin, double yaw, double pitch, double fov) {
    s->camera.vw = vw; s->camera.vh = vh; s->camera.origin = V(origin); s->camera.yaw = yaw; s->camera.pitch = pitch; s->camera.set_fov(fov);
}
void rth_set_resolution(Scene* s, int32_t vw, int32_t vh) { s->camera.vw = vw; s->camera.vh = vh; }
void rth_set_options(Scene* s, uint32_t max_ray_depth, const uint8_t* ambient) { s->options.max_ray_depth = max_ray_depth; s->options.ambient = C(ambient); }

int rth_add_sphere(Scene* s, const double* pos, double radius, const rth_material* m) {
    SceneObject o; o.kind = SceneObject::KSphere; o.sphere.origin = V(pos); o.sphere.radius = radius; o.sphere.material = M(m);
    s->objects.push_back(std::move(o)); return (int)s->objects.size() - 1;
}
int rth_add_plane(Scene* s, const double* origin, const double* normal, float uv_wrap, const rth_material* m) {
    SceneObject o; o.kind = SceneObject::KPlane; o.plane.origin = V(origin); o.plane.normal = V(normal); o.plane.uv_wrap = uv_wrap; o.plane.material = M(m);
    s->objects.push_back(std::move(o)); return (int)s->objects.size() - 1;
}
int rth_add_aabb(Scene* s, const double* pos, const double* size, const rth_material* m) {
    SceneObject o; o.kind = SceneObject::KAabb; o.aabb.pos = V(pos); o.aabb.size = V(size); o.aabb.material = M(m);
    s->objects.push_back(std::move(o)); return (int)s->objects.size() - 1;
}

// ---- meshes
std::shared_ptr<Mesh>* rth_mesh_new() { return new std::shared_ptr<Mesh>(std::make_shared<Mesh>()); }
void rth_mesh_free(std::shared_ptr<Mesh>* m) { delete m; }
std::shared_ptr<Mesh>* rth_mesh_from_obj(const char* path) {
    auto m = std::make_shared<Mesh>();
    std::string err;
    if (!Mesh::from_obj(path, Material(), *m, err)) { g_err = err; return nullptr; }
    return new std::shared_ptr<Mesh>(m);
}
std::shared_ptr<Mesh>* rth_mesh_heightfield(uint32_t nx, uint32_t nz, double cell, double amplitude, double base_freq, uint32_t octaves, uint64_t seed) {
    return new std::shared_ptr<Mesh>(make_heightfield(nx, nz, cell, amplitude, base_freq, octaves, seed));
}
int rth_mesh_set_geometry(std::shared_ptr<Mesh>* mp, uint32_t n_verts, const double* verts, uint32_t n_tris, const uint32_t* tris) {
    Mesh& m = **mp;
    m.verts.resize(n_verts);
    for (uint32_t i = 0; i < n_verts; i++) m.verts[i] = V(verts + 3 * i);
    m.tris.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        m.tris[i] = {tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]};
        for (int c = 0; c < 3; c++) if (tris[3 * i + c] >= n_verts) return fail("mesh: vertex index out of range");
    }
    return 0;
}
int rth_mesh_set_normals(std::shared_ptr<Mesh>* mp, uint32_t n_normals, const double* normals, uint32_t n_tris, const uint32_t* tri_normals) {
    Mesh& m = **mp;
    if (n_tris != m.tris.size()) return fail("mesh: normal index count differs from triangle count");
    m.normals.resize(n_normals);
    for (uint32_t i = 0; i < n_normals; i++) m.normals[i] = V(normals + 3 * i);
    m.tri_normals.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        m.tri_normals[i] = {tri_normals[3 * i], tri_normals[3 * i + 1], tri_normals[3 * i + 2]};
        for (int c = 0; c < 3; c++) if (tri_normals[3 * i + c] >= n_normals) return fail("mesh: normal index out of range");
    }
    return 0;
}
int rth_mesh_set_texcoords(std::shared_ptr<Mesh>* mp, uint32_t n_tc, const float* tc, uint32_t n_tris, const uint32_t* tri_tc) {
    Mesh& m = **mp;
    if (n_tris != m.tris.size()) return fail("mesh: texcoord index count differs from triangle count");
    m.texcoords.resize(n_tc);
    for (uint32_t i = 0; i < n_tc; i++) m.texcoords[i] = {tc[2 * i], tc[2 * i + 1]};
    m.tri_texcoords.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        m.tri_texcoords[i] = {tri_tc[3 * i], tri_tc[3 * i + 1], tri_tc[3 * i + 2]};
        for (int c = 0; c < 3; c++) if (tri_tc[3 * i + c] >= n_tc) return fail("mesh: texcoord index out of range");
    }
    return 0;
}
void rth_mesh_scale(std::shared_ptr<Mesh>* m, double d) { (*m)->scale(d); }
void rth_mesh_center(std::shared_ptr<Mesh>* m) { (*m)->center(); }
void rth_mesh_shift(std::shared_ptr<Mesh>* m, const double* d) { (*m)->shift(V(d)); }
void rth_mesh_rotate_xyz(std::shared_ptr<Mesh>* m, const double* r) { (*m)->rotate_xyz(V(r)); }
void rth_mesh_rotate_zyx(std::shared_ptr<Mesh>* m, const double* r) { (*m)->rotate_zyx(V(r)); }
void rth_mesh_recalculate_normals(std::shared_ptr<Mesh>* m) { (*m)->recalculate_normals(); }
int rth_mesh_generate_sbvh(std::shared_ptr<Mesh>* m) { std::string e; return (*m)->generate_sbvh(e) ? 0 : fail(e); }
int rth_mesh_set_verts(std::shared_ptr<Mesh>* m, uint32_t n, const double* verts) {
    if (!verts || n != (*m)->verts.size()) return fail("set_verts: the number of vertices must not change");
    for (uint32_t i = 0; i < n; i++) (*m)->verts[i] = Vector3{verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
    return 0;
}
// The tree is left to the render library: the flattened mesh carries no nodes and rt_scene_upload builds them on the device.
void rth_mesh_use_device_bvh(std::shared_ptr<Mesh>* m, int on) { (*m)->device_bvh = on != 0; }
// Mesh::sbvh is a public field of the reference (mesh.rs:118): a caller may install a tree of their own.  The nodes
// are taken as given (rt_bvh_node layout: a node's first child is the next record); upload validates them.
int rth_mesh_set_bvh(std::shared_ptr<Mesh>* m, const rt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* prim_indices, uint32_t n_prims, uint32_t depth) {
    if (!nodes || !prim_indices || n_nodes == 0 || n_prims == 0) return fail("set_bvh: empty tree");
    auto b = std::make_shared<FlatBvh>();
    b->nodes.assign(nodes, nodes + n_nodes);
    b->prim_indices.assign(prim_indices, prim_indices + n_prims);
    b->depth = depth;
    (*m)->sbvh = b;
    return 0;
}
uint32_t rth_mesh_n_tris(std::shared_ptr<Mesh>* m) { return (uint32_t)(*m)->tris.size(); }
uint32_t rth_mesh_n_verts(std::shared_ptr<Mesh>* m) { return (uint32_t)(*m)->verts.size(); }
uint32_t rth_mesh_n_normals(std::shared_ptr<Mesh>* m) { return (uint32_t)(*m)->normals.size(); }
uint32_t rth_mesh_n_nodes(std::shared_ptr<Mesh>* m) { return (*m)->sbvh ? (uint32_t)(*m)->sbvh->nodes.size() : 0; }
uint32_t rth_mesh_bvh_depth(std::shared_ptr<Mesh>* m) { return (*m)->sbvh ? (*m)->sbvh->depth : 0; }
const double* rth_mesh_verts(std::shared_ptr<Mesh>* m) { return reinterpret_cast<const double*>((*m)->verts.data()); }
const double* rth_mesh_normals(std::shared_ptr<Mesh>* m) { return reinterpret_cast<const double*>((*m)->normals.data()); }
const uint32_t* rth_mesh_tri_normals(std::shared_ptr<Mesh>* m) { return reinterpret_cast<const uint32_t*>((*m)->tri_normals.data()); }
const uint32_t* rth_mesh_tris(std::shared_ptr<Mesh>* m) { return reinterpret_cast<const uint32_t*>((*m)->tris.data()); }
int rth_add_mesh(Scene* s, std::shared_ptr<Mesh>* m, const rth_material* mat) {
    // each scene object owns its material; the geometry may be shared
    auto copy = std::make_shared<Mesh>(**m);
    copy->material = M(mat);
    SceneObject o; o.kind = SceneObject::KMesh; o.mesh = copy;
    s->objects.push_back(std::move(o)); return (int)s->objects.size() - 1;
}

// A handle on the mesh object `index` of the scene holds (rth_add_mesh copied the caller's): changes through it
// (rth_mesh_set_verts, rth_mesh_recalculate_normals ...) show in the scene's next flatten.  NULL if it is no mesh.
std::shared_ptr<Mesh>* rth_scene_object_mesh(Scene* s, uint32_t index) {
    if (index >= s->objects.size() || s->objects[index].kind != SceneObject::KMesh) { g_err = "object is not a mesh"; return nullptr; }
    return new std::shared_ptr<Mesh>(s->objects[index].mesh);
}

// ---- lights
int rth_add_sun(Scene* s, const uint8_t* color, double intensity, int32_t specular_power, double specular_strength,
                const double* vector, int shadows, double shadow_coefficient) {
    Light l = Light::sun();
    l.color = C(color); l.intensity = intensity; l.specular_power = specular_power; l.specular_strength = specular_strength;
    l.vector = V(vector); l.shadows = shadows != 0; l.shadow_coefficient = shadow_coefficient;
    s->lights.push_back(l); return (int)s->lights.size() - 1;
}
int rth_add_point(Scene* s, const uint8_t* color, double intensity, int32_t specular_power, double specular_strength,
                  const double* position, double max_distance) {
    Light l = Light::point();
    l.color = C(color); l.intensity = intensity; l.specular_power = specular_power; l.specular_strength = specular_strength;
    l.vector = V(position); l.max_distance = max_distance;
    s->lights.push_back(l); return (int)s->lights.size() - 1;
}
int rth_add_area(Scene* s, const uint8_t* color, double intensity, int32_t specular_power, double specular_strength,
                 uint32_t surface, const double* centre, double radius, const double* corners, uint32_t iterations, double max_distance) {
    Light l = Light::area();
    l.color = C(color); l.intensity = intensity; l.specular_power = specular_power; l.specular_strength = specular_strength;
    l.area_surface = surface; l.area_iterations = iterations; l.max_distance = max_distance; l.area_radius = radius;
    if (centre) l.vector = V(centre);
    if (corners) for (int c = 0; c < 4; c++) l.area_corners[c] = V(corners + 3 * c);
    s->lights.push_back(l); return (int)s->lights.size() - 1;
}

// ---- skybox
void rth_set_skybox_solid(Scene* s, const uint8_t* color) { s->skybox = Skybox(); s->skybox.kind = Skybox::Solid; s->skybox.color = C(color); }
void rth_set_skybox_normal(Scene* s) { s->skybox = Skybox(); s->skybox.kind = Skybox::Normal; }
int rth_set_skybox_cubemap(Scene* s, uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (w / 4 != h / 3 || w < 4) return fail("cubemap: width/4 must equal height/3 (skybox.rs:49-51)");
    s->skybox = Skybox(); s->skybox.kind = Skybox::Cubemap; s->skybox.w = w; s->skybox.h = h;
    s->skybox.rgb = std::make_shared<std::vector<uint8_t>>(rgb, rgb + (size_t)w * h * 3);
    return 0;
}
int rth_set_skybox_star_atlas(Scene* s, uint32_t cell, uint64_t seed) {
    s->skybox = Skybox(); s->skybox.kind = Skybox::Cubemap; s->skybox.w = 4 * cell; s->skybox.h = 3 * cell;
    s->skybox.rgb = std::make_shared<std::vector<uint8_t>>();
    make_star_atlas(cell, seed, *s->skybox.rgb);
    return 0;
}

// ---- flatten / io / render
const rt_scene_desc* rth_scene_flatten(Scene* s) {
    std::string e; const rt_scene_desc* d = s->flatten(e);
    if (!d) g_err = e;
    return d;
}
int rth_scene_save(Scene* s, const char* path) { std::string e; return s->save(path, e) ? 0 : fail(e); }
LoadedDesc* rth_desc_load(const char* path) { std::string e; LoadedDesc* l = load_desc(path, e); if (!l) g_err = e; return l; }
const rt_scene_desc* rth_desc_get(LoadedDesc* l) { return desc_of(l); }
void rth_desc_free(LoadedDesc* l) { free_desc(l); }
// Sequences: marks every mesh of a flattened scene as unchanged since the context's last upload (rt_mesh::unchanged).
// (keep: 0 uploaded afresh, 1 unchanged, 2 same connectivity and tree with new vertex positions and normals: refitted)
void rth_desc_keep_meshes(const rt_scene_desc* d, int keep) {
    rt_mesh* m = const_cast<rt_mesh*>(d->meshes);
    for (uint32_t i = 0; i < d->n_meshes; i++) m[i].unchanged = (uint32_t)(keep < 0 ? 0 : keep > 2 ? 2 : keep);
}
void rth_desc_set_resolution(LoadedDesc* l, int32_t vw, int32_t vh) {
    rt_scene_desc* d = const_cast<rt_scene_desc*>(desc_of(l)); d->camera.vw = vw; d->camera.vh = vh;
}
// Points a loaded scene's skybox at a caller-owned 4x3 cubemap atlas (kept alive by the caller while the scene is used)
int rth_desc_set_cubemap(LoadedDesc* l, uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (!rgb || w < 4 || w / 4 != h / 3) { g_err = "cubemap: width/4 must equal height/3 (skybox.rs:49-51)"; return -1; }
    rt_scene_desc* d = const_cast<rt_scene_desc*>(desc_of(l));
    d->skybox.kind = RT_SKY_CUBEMAP; d->skybox.w = w; d->skybox.h = h; d->skybox.rgb = rgb;
    return 0;
}
void rth_desc_info(const rt_scene_desc* d, uint32_t* out) {   // vw, vh, n_objects, n_lights, n_meshes, total tris, total nodes, skybox kind
    out[0] = (uint32_t)d->camera.vw; out[1] = (uint32_t)d->camera.vh; out[2] = d->n_objects; out[3] = d->n_lights; out[4] = d->n_meshes;
    uint32_t t = 0, n = 0;
    for (uint32_t i = 0; i < d->n_meshes; i++) { t += d->meshes[i].n_tris; n += d->meshes[i].n_nodes; }
    out[5] = t; out[6] = n; out[7] = d->skybox.kind;
}

// Scene::render (scene.rs:220-246): rgb_out = vw*vh*3 bytes
int rth_scene_render(Scene* s, uint8_t* rgb_out, int device_id) {
    std::vector<Color> px; std::string e;
    if (!s->render(px, e, device_id)) return fail(e);
    std::memcpy(rgb_out, px.data(), px.size() * 3);
    return 0;
}
// Scene::render_to (scene.rs:249-265)
int rth_scene_render_to(Scene* s, const char* path, int device_id) { std::string e; return s->render_to(path, e, device_id) ? 0 : fail(e); }

int rth_write_png(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb) { std::string e; return write_png(path, w, h, rgb, e) ? 0 : fail(e); }
// two-call protocol: first with rgb == NULL to get the size
int rth_read_png(const char* path, uint32_t* w, uint32_t* h, uint8_t* rgb) {
    std::vector<uint8_t> px; std::string e;
    if (!read_png(path, *w, *h, px, e)) return fail(e);
    if (rgb) std::memcpy(rgb, px.data(), px.size());
    return 0;
}
// PNG or baseline JPEG, told apart by content; same two-call protocol
int rth_read_image(const char* path, uint32_t* w, uint32_t* h, uint8_t* rgb) {
    std::vector<uint8_t> px; std::string e;
    if (!read_image(path, *w, *h, px, e)) return fail(e);
    if (rgb) std::memcpy(rgb, px.data(), px.size());
    return 0;
}

// The scene description language (sdl/src/main.rs:49-53: Interpreter::new(file).run()).  source_or_path is the script
// text (is_path == 0) or a file name; t is the global `t`; seed feeds random().  On success returns a Scene (free it
// with rth_scene_free) and, through rth_sdl_printed(), what print() wrote during this call.
static thread_local std::string g_printed;
Scene* rth_sdl_load(const char* source_or_path, int is_path, const char* base_dir, double t, uint64_t seed) {
    SdlOptions opt; opt.t = t; opt.seed = seed; opt.base_dir = base_dir ? base_dir : "";
    Scene* s = new Scene();
    std::string err;
    g_printed.clear();
    const bool ok = is_path ? load_sdl_file(source_or_path, opt, *s, g_printed, err) : load_sdl(source_or_path, opt, *s, g_printed, err);
    if (!ok) { delete s; g_err = err; return nullptr; }
    return s;
}
const char* rth_sdl_printed() { return g_printed.c_str(); }
void rth_scene_counts(Scene* s, uint32_t* out) { out[0] = (uint32_t)s->objects.size(); out[1] = (uint32_t)s->lights.size(); out[2] = (uint32_t)s->camera.vw; out[3] = (uint32_t)s->camera.vh; }
// Camera::chf (camera.rs:45-47), for callers that fill an rt_camera themselves
double rth_camera_chf(double fov) { return Camera::chf_of(fov); }

}  // extern "C"
