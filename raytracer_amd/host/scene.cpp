// Scene flattening, (de)serialisation, the render()/render_to() entry points
// and the seeded synthetic generators.
#include <zlib.h>

#include <array>
#include <cstdio>
#include <cstring>
#include <map>

#include "scene.hpp"

namespace rth {

Texture Texture::image(uint32_t w, uint32_t h, const uint8_t* rgb) {
    Texture t; t.kind = Image; t.w = w; t.h = h;
    t.rgb = std::make_shared<std::vector<uint8_t>>(rgb, rgb + (size_t)w * h * 3);
    return t;
}

struct Scene::Flat {
    rt_scene_desc desc;
    std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_box> boxes;
    std::vector<rt_mesh> meshes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
};

Scene::~Scene() {
    delete flat_;
    if (ctx_) rt_ctx_destroy(ctx_);
}

static void put3(double* d, Vector3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

const rt_scene_desc* Scene::flatten(std::string& err) {
    delete flat_;
    flat_ = new Flat();
    Flat& f = *flat_;
    std::memset(&f.desc, 0, sizeof(f.desc));
    f.desc.camera.vw = camera.vw; f.desc.camera.vh = camera.vh;
    put3(f.desc.camera.origin, camera.origin);
    f.desc.camera.yaw = camera.yaw; f.desc.camera.pitch = camera.pitch;
    f.desc.camera.fov = camera.fov; f.desc.camera.chf = camera.chf;
    f.desc.options.max_ray_depth = options.max_ray_depth;
    f.desc.options.ambient[0] = options.ambient.r; f.desc.options.ambient[1] = options.ambient.g; f.desc.options.ambient[2] = options.ambient.b;

    std::map<const Mesh*, uint32_t> mesh_ids;
    for (const SceneObject& o : objects) {
        const Material& m = o.material();
        rt_texture t; std::memset(&t, 0, sizeof(t));
        t.kind = (uint32_t)m.texture.kind;
        t.a[0] = m.texture.a.r; t.a[1] = m.texture.a.g; t.a[2] = m.texture.a.b;
        t.b[0] = m.texture.b.r; t.b[1] = m.texture.b.g; t.b[2] = m.texture.b.b;
        if (m.texture.kind == Texture::Image) {
            if (!m.texture.rgb || m.texture.rgb->size() != (size_t)m.texture.w * m.texture.h * 3 || m.texture.w == 0) { err = "image texture without pixels"; return nullptr; }
            t.w = m.texture.w; t.h = m.texture.h; t.rgb = m.texture.rgb->data();
        }
        f.textures.push_back(t);
        rt_material rm; std::memset(&rm, 0, sizeof(rm));
        rm.texture = (uint32_t)f.textures.size() - 1;
        rm.reflectiveness = m.reflectiveness; rm.transparency = m.transparency; rm.ior = m.ior; rm.emissivity = m.emissivity;
        f.materials.push_back(rm);

        rt_object ro; std::memset(&ro, 0, sizeof(ro));
        ro.kind = (uint32_t)o.kind; ro.material = (uint32_t)f.materials.size() - 1;
        switch (o.kind) {
            case SceneObject::KSphere: {
                rt_sphere s; put3(s.c, o.sphere.origin); s.r = o.sphere.radius;
                ro.payload = (uint32_t)f.spheres.size(); f.spheres.push_back(s); break;
            }
            case SceneObject::KPlane: {
                rt_plane p; std::memset(&p, 0, sizeof(p));
                put3(p.p0, o.plane.origin); put3(p.n, o.plane.normal); p.uv_wrap = o.plane.uv_wrap;
                ro.payload = (uint32_t)f.planes.size(); f.planes.push_back(p); break;
            }
            case SceneObject::KAabb: {
                rt_box b; put3(b.pos, o.aabb.pos); put3(b.half, o.aabb.size);
                ro.payload = (uint32_t)f.boxes.size(); f.boxes.push_back(b); break;
            }
            case SceneObject::KMesh: {
                const Mesh* mp = o.mesh.get();
                if (!mp || (!mp->sbvh && !mp->device_bvh)) { err = "mesh without an acceleration structure (mesh.rs:350 asserts sbvh.is_some())"; return nullptr; }
                auto it = mesh_ids.find(mp);
                if (it == mesh_ids.end()) {
                    if (mp->tri_normals.size() != mp->tris.size()) { err = "mesh: tri_normals and tris differ in length"; return nullptr; }
                    if (!mp->tri_texcoords.empty() && mp->tri_texcoords.size() != mp->tris.size()) { err = "mesh: tri_texcoords and tris differ in length"; return nullptr; }
                    rt_mesh rm2; std::memset(&rm2, 0, sizeof(rm2));
                    rm2.verts = reinterpret_cast<const double*>(mp->verts.data()); rm2.n_verts = (uint32_t)mp->verts.size();
                    rm2.normals = reinterpret_cast<const double*>(mp->normals.data()); rm2.n_normals = (uint32_t)mp->normals.size();
                    rm2.texcoords = reinterpret_cast<const float*>(mp->texcoords.data()); rm2.n_texcoords = (uint32_t)mp->texcoords.size();
                    rm2.tri_pos = reinterpret_cast<const uint32_t*>(mp->tris.data());
                    rm2.tri_nrm = reinterpret_cast<const uint32_t*>(mp->tri_normals.data());
                    rm2.tri_uv = mp->tri_texcoords.empty() ? nullptr : reinterpret_cast<const uint32_t*>(mp->tri_texcoords.data());
                    rm2.n_tris = (uint32_t)mp->tris.size();
                    if (!mp->device_bvh) {
                        rm2.nodes = mp->sbvh->nodes.data(); rm2.n_nodes = (uint32_t)mp->sbvh->nodes.size();
                        rm2.prim_indices = mp->sbvh->prim_indices.data(); rm2.n_prims = (uint32_t)mp->sbvh->prim_indices.size();
                    }
                    it = mesh_ids.emplace(mp, (uint32_t)f.meshes.size()).first;
                    f.meshes.push_back(rm2);
                }
                ro.payload = it->second; break;
            }
        }
        f.objects.push_back(ro);
    }
    for (const Light& l : lights) {
        rt_light rl; std::memset(&rl, 0, sizeof(rl));
        rl.kind = (uint32_t)l.kind;
        rl.color[0] = l.color.r; rl.color[1] = l.color.g; rl.color[2] = l.color.b;
        rl.shadows = l.shadows ? 1 : 0;
        rl.intensity = l.intensity; rl.specular_power = l.specular_power; rl.specular_strength = l.specular_strength;
        put3(rl.v, l.vector);
        rl.shadow_coefficient = l.shadow_coefficient; rl.max_distance = l.max_distance;
        rl.area_surface = l.area_surface; rl.area_iterations = l.area_iterations; rl.area_radius = l.area_radius;
        for (int c = 0; c < 4; c++) put3(rl.area_corners[c], l.area_corners[c]);
        f.lights.push_back(rl);
    }
    f.desc.skybox.kind = (uint32_t)skybox.kind;
    f.desc.skybox.color[0] = skybox.color.r; f.desc.skybox.color[1] = skybox.color.g; f.desc.skybox.color[2] = skybox.color.b;
    if (skybox.kind == Skybox::Cubemap) {
        if (!skybox.rgb || skybox.rgb->size() != (size_t)skybox.w * skybox.h * 3) { err = "cubemap without pixels"; return nullptr; }
        if (skybox.w / 4 != skybox.h / 3 || skybox.w < 4) { err = "cubemap: width/4 must equal height/3 (skybox.rs:49-51)"; return nullptr; }
        f.desc.skybox.w = skybox.w; f.desc.skybox.h = skybox.h; f.desc.skybox.rgb = skybox.rgb->data();
    }
    f.desc.objects = f.objects.data(); f.desc.n_objects = (uint32_t)f.objects.size();
    f.desc.spheres = f.spheres.data(); f.desc.n_spheres = (uint32_t)f.spheres.size();
    f.desc.planes = f.planes.data(); f.desc.n_planes = (uint32_t)f.planes.size();
    f.desc.boxes = f.boxes.data(); f.desc.n_boxes = (uint32_t)f.boxes.size();
    f.desc.meshes = f.meshes.data(); f.desc.n_meshes = (uint32_t)f.meshes.size();
    f.desc.materials = f.materials.data(); f.desc.n_materials = (uint32_t)f.materials.size();
    f.desc.textures = f.textures.data(); f.desc.n_textures = (uint32_t)f.textures.size();
    f.desc.lights = f.lights.data(); f.desc.n_lights = (uint32_t)f.lights.size();
    return &f.desc;
}

// scene.rs:220-246
bool Scene::render(std::vector<Color>& out, std::string& err, int device_id) {
    const rt_scene_desc* d = flatten(err);
    if (!d) return false;
    if (!ctx_) {
        int rc = rt_ctx_create(device_id, &ctx_);
        if (rc != RT_OK) { err = std::string("rt_ctx_create: ") + rt_last_error(nullptr); ctx_ = nullptr; return false; }
    }
    int rc = rt_scene_upload(ctx_, d);
    if (rc != RT_OK) { err = std::string("rt_scene_upload: ") + rt_last_error(ctx_); return false; }
    std::vector<uint8_t> rgb((size_t)camera.vw * camera.vh * 3);
    rc = rt_render(ctx_, rgb.data(), nullptr, nullptr);
    if (rc != RT_OK) { err = std::string("rt_render: ") + rt_last_error(ctx_); return false; }
    out.resize((size_t)camera.vw * camera.vh);
    for (size_t i = 0; i < out.size(); i++) out[i] = Color(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return true;
}

// scene.rs:249-265
bool Scene::render_to(const std::string& path, std::string& err, int device_id) {
    std::vector<Color> px;
    if (!render(px, err, device_id)) return false;
    static_assert(sizeof(Color) == 3, "Color must be 3 packed bytes");
    return write_png(path, (uint32_t)camera.vw, (uint32_t)camera.vh, reinterpret_cast<const uint8_t*>(px.data()), err);
}

// ------------------------------------------------------------------ generators
namespace {
inline uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double lattice(int64_t ix, int64_t iz, uint64_t seed) {
    uint64_t h = mix64(seed ^ mix64((uint64_t)ix * 0x9E3779B97F4A7C15ull ^ mix64((uint64_t)iz)));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}
inline double value_noise(double x, double z, uint64_t seed) {
    double fx = std::floor(x), fz = std::floor(z);
    int64_t ix = (int64_t)fx, iz = (int64_t)fz;
    double tx = x - fx, tz = z - fz;
    double sx = tx * tx * (3. - 2. * tx), sz = tz * tz * (3. - 2. * tz);
    double a = lattice(ix, iz, seed), b = lattice(ix + 1, iz, seed);
    double c = lattice(ix, iz + 1, seed), d = lattice(ix + 1, iz + 1, seed);
    double ab = a + (b - a) * sx, cd = c + (d - c) * sx;
    return ab + (cd - ab) * sz;
}
}  // namespace

// A (nx x nz)-cell height field with the triangle pattern of scenes/terrain.sdl:55-65:
// 2*nx*nz triangles over (nx+1)*(nz+1) vertices, y = fBm value noise.
std::shared_ptr<Mesh> make_heightfield(uint32_t nx, uint32_t nz, double cell, double amplitude, double base_freq,
                                       uint32_t octaves, uint64_t seed) {
    auto m = std::make_shared<Mesh>();
    const uint32_t w = nx + 1, h = nz + 1;
    m->verts.reserve((size_t)w * h);
    for (uint32_t z = 0; z < h; z++)
        for (uint32_t x = 0; x < w; x++) {
            double v = 0., freq = base_freq, amp = 1.;
            for (uint32_t o = 0; o < octaves; o++) {
                v = v + value_noise((double)x * freq, (double)z * freq, seed + o) * amp;
                freq = freq * 2.; amp = amp * 0.5;
            }
            m->verts.emplace_back((double)x * cell, v * amplitude, (double)z * cell);
        }
    m->tris.reserve((size_t)nx * nz * 2);
    for (uint32_t z = 0; z < nz; z++)
        for (uint32_t x = 0; x < nx; x++) {
            m->tris.push_back({x + z * w, x + (z + 1) * w, x + 1 + z * w});
            m->tris.push_back({x + (z + 1) * w, x + 1 + (z + 1) * w, x + 1 + z * w});
        }
    return m;
}

// Procedural 4x3 cubemap atlas (cell x cell faces): a vertical gradient plus
// seeded "stars", so that neighbouring texels differ and bilinear taps matter.
void make_star_atlas(uint32_t cell, uint64_t seed, std::vector<uint8_t>& rgb) {
    const uint32_t w = 4 * cell, h = 3 * cell;
    rgb.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint64_t hsh = mix64(seed ^ mix64(((uint64_t)y << 32) | x));
            uint8_t* p = &rgb[3 * ((size_t)y * w + x)];
            uint32_t cx = x / cell, cy = y / cell;
            uint8_t base_r = (uint8_t)(20 + 50 * cx + (y % cell) * 40 / cell);
            uint8_t base_g = (uint8_t)(30 + 60 * cy + (x % cell) * 50 / cell);
            uint8_t base_b = (uint8_t)(90 + (hsh >> 40) % 60);
            if ((hsh & 0xFF) < 6) { p[0] = 255; p[1] = (uint8_t)(200 + (hsh >> 8) % 56); p[2] = (uint8_t)(180 + (hsh >> 16) % 76); }
            else { p[0] = base_r; p[1] = base_g; p[2] = base_b; }
        }
}

// ------------------------------------------------------------------ (de)serialisation
namespace {
struct GzW {
    gzFile f; bool ok = true;
    void raw(const void* p, size_t n) { while (ok && n) { unsigned c = (unsigned)std::min<size_t>(n, 1u << 30); if (gzwrite(f, p, c) != (int)c) ok = false; p = (const char*)p + c; n -= c; } }
    void u32(uint32_t v) { raw(&v, 4); }
};
struct GzR {
    gzFile f; bool ok = true;
    void raw(void* p, size_t n) { while (ok && n) { unsigned c = (unsigned)std::min<size_t>(n, 1u << 30); if (gzread(f, p, c) != (int)c) ok = false; p = (char*)p + c; n -= c; } }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
};
const char MAGIC[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', '1'};
}  // namespace

bool save_desc(const rt_scene_desc* d, const std::string& path, std::string& err) {
    gzFile f = gzopen(path.c_str(), "wb6");
    if (!f) { err = "cannot open " + path; return false; }
    GzW w{f};
    w.raw(MAGIC, 8);
    w.raw(&d->camera, sizeof(d->camera)); w.raw(&d->options, sizeof(d->options));
    w.u32(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; i++) {
        rt_texture t = d->textures[i]; const uint8_t* px = t.rgb; t.rgb = nullptr;
        w.raw(&t, sizeof(t));
        if (t.kind == RT_TEX_IMAGE) w.raw(px, (size_t)t.w * t.h * 3);
    }
    w.u32(d->n_materials); w.raw(d->materials, sizeof(rt_material) * d->n_materials);
    w.u32(d->n_objects); w.raw(d->objects, sizeof(rt_object) * d->n_objects);
    w.u32(d->n_spheres); w.raw(d->spheres, sizeof(rt_sphere) * d->n_spheres);
    w.u32(d->n_planes); w.raw(d->planes, sizeof(rt_plane) * d->n_planes);
    w.u32(d->n_boxes); w.raw(d->boxes, sizeof(rt_box) * d->n_boxes);
    w.u32(d->n_meshes);
    for (uint32_t i = 0; i < d->n_meshes; i++) {
        const rt_mesh& m = d->meshes[i];
        w.u32(m.n_verts); w.raw(m.verts, sizeof(double) * 3 * m.n_verts);
        w.u32(m.n_normals); w.raw(m.normals, sizeof(double) * 3 * m.n_normals);
        w.u32(m.n_texcoords); w.raw(m.texcoords, sizeof(float) * 2 * m.n_texcoords);
        w.u32(m.n_tris); w.raw(m.tri_pos, 12 * (size_t)m.n_tris); w.raw(m.tri_nrm, 12 * (size_t)m.n_tris);
        w.u32(m.tri_uv ? 1 : 0); if (m.tri_uv) w.raw(m.tri_uv, 12 * (size_t)m.n_tris);
        w.u32(m.n_nodes); w.raw(m.nodes, sizeof(rt_bvh_node) * m.n_nodes);
        w.u32(m.n_prims); w.raw(m.prim_indices, 4 * (size_t)m.n_prims);
    }
    w.u32(d->n_lights); w.raw(d->lights, sizeof(rt_light) * d->n_lights);
    rt_skybox sb = d->skybox; const uint8_t* px = sb.rgb; sb.rgb = nullptr;
    w.raw(&sb, sizeof(sb));
    if (sb.kind == RT_SKY_CUBEMAP) w.raw(px, (size_t)sb.w * sb.h * 3);
    bool ok = w.ok;
    if (gzclose(f) != Z_OK) ok = false;
    if (!ok) err = "write failed: " + path;
    return ok;
}

struct LoadedDesc {
    rt_scene_desc desc;
    std::vector<rt_texture> textures; std::vector<std::vector<uint8_t>> blobs;
    std::vector<rt_material> materials; std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres; std::vector<rt_plane> planes; std::vector<rt_box> boxes;
    std::vector<rt_mesh> meshes; std::vector<rt_light> lights;
    std::vector<std::vector<double>> dbl; std::vector<std::vector<float>> flt; std::vector<std::vector<uint32_t>> u32s;
    std::vector<std::vector<rt_bvh_node>> nodes;
};

LoadedDesc* load_desc(const std::string& path, std::string& err) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return nullptr; }
    GzR r{f};
    auto L = std::make_unique<LoadedDesc>();
    std::memset(&L->desc, 0, sizeof(L->desc));
    char magic[8]; r.raw(magic, 8);
    if (!r.ok || std::memcmp(magic, MAGIC, 8) != 0) { gzclose(f); err = "not a scene file: " + path; return nullptr; }
    const uint32_t LIMIT = 1u << 28;
    bool bad = false;
    auto cnt = [&]() { uint32_t n = r.u32(); if (n > LIMIT) { bad = true; r.ok = false; n = 0; } return n; };
    r.raw(&L->desc.camera, sizeof(rt_camera)); r.raw(&L->desc.options, sizeof(rt_options));
    uint32_t nt = cnt(); L->textures.resize(nt);
    for (uint32_t i = 0; i < nt && r.ok; i++) {
        rt_texture& t = L->textures[i]; r.raw(&t, sizeof(t)); t.rgb = nullptr;
        if (t.kind == RT_TEX_IMAGE) {
            if ((uint64_t)t.w * t.h > LIMIT) { bad = true; break; }
            L->blobs.emplace_back((size_t)t.w * t.h * 3); r.raw(L->blobs.back().data(), L->blobs.back().size()); t.rgb = L->blobs.back().data();
        }
    }
    uint32_t n;
    n = cnt(); L->materials.resize(n); r.raw(L->materials.data(), sizeof(rt_material) * n);
    n = cnt(); L->objects.resize(n); r.raw(L->objects.data(), sizeof(rt_object) * n);
    n = cnt(); L->spheres.resize(n); r.raw(L->spheres.data(), sizeof(rt_sphere) * n);
    n = cnt(); L->planes.resize(n); r.raw(L->planes.data(), sizeof(rt_plane) * n);
    n = cnt(); L->boxes.resize(n); r.raw(L->boxes.data(), sizeof(rt_box) * n);
    uint32_t nm = cnt(); L->meshes.resize(nm);
    for (uint32_t i = 0; i < nm && r.ok; i++) {
        rt_mesh& m = L->meshes[i]; std::memset(&m, 0, sizeof(m));
        m.n_verts = cnt(); L->dbl.emplace_back((size_t)m.n_verts * 3); r.raw(L->dbl.back().data(), 24 * (size_t)m.n_verts); m.verts = L->dbl.back().data();
        m.n_normals = cnt(); L->dbl.emplace_back((size_t)m.n_normals * 3); r.raw(L->dbl.back().data(), 24 * (size_t)m.n_normals); m.normals = L->dbl.back().data();
        m.n_texcoords = cnt(); L->flt.emplace_back((size_t)m.n_texcoords * 2); r.raw(L->flt.back().data(), 8 * (size_t)m.n_texcoords); m.texcoords = L->flt.back().data();
        m.n_tris = cnt();
        L->u32s.emplace_back((size_t)m.n_tris * 3); r.raw(L->u32s.back().data(), 12 * (size_t)m.n_tris); m.tri_pos = L->u32s.back().data();
        L->u32s.emplace_back((size_t)m.n_tris * 3); r.raw(L->u32s.back().data(), 12 * (size_t)m.n_tris); m.tri_nrm = L->u32s.back().data();
        if (r.u32()) { L->u32s.emplace_back((size_t)m.n_tris * 3); r.raw(L->u32s.back().data(), 12 * (size_t)m.n_tris); m.tri_uv = L->u32s.back().data(); }
        m.n_nodes = cnt(); L->nodes.emplace_back((size_t)m.n_nodes); r.raw(L->nodes.back().data(), sizeof(rt_bvh_node) * m.n_nodes); m.nodes = L->nodes.back().data();
        m.n_prims = cnt(); L->u32s.emplace_back((size_t)m.n_prims); r.raw(L->u32s.back().data(), 4 * (size_t)m.n_prims); m.prim_indices = L->u32s.back().data();
    }
    n = cnt(); L->lights.resize(n); r.raw(L->lights.data(), sizeof(rt_light) * n);
    r.raw(&L->desc.skybox, sizeof(rt_skybox)); L->desc.skybox.rgb = nullptr;
    if (r.ok && L->desc.skybox.kind == RT_SKY_CUBEMAP) {
        if ((uint64_t)L->desc.skybox.w * L->desc.skybox.h > LIMIT) bad = true;
        else { L->blobs.emplace_back((size_t)L->desc.skybox.w * L->desc.skybox.h * 3); r.raw(L->blobs.back().data(), L->blobs.back().size()); L->desc.skybox.rgb = L->blobs.back().data(); }
    }
    gzclose(f);
    if (!r.ok || bad) { err = "truncated or corrupt scene file: " + path; return nullptr; }
    rt_scene_desc& d = L->desc;
    d.textures = L->textures.data(); d.n_textures = (uint32_t)L->textures.size();
    d.materials = L->materials.data(); d.n_materials = (uint32_t)L->materials.size();
    d.objects = L->objects.data(); d.n_objects = (uint32_t)L->objects.size();
    d.spheres = L->spheres.data(); d.n_spheres = (uint32_t)L->spheres.size();
    d.planes = L->planes.data(); d.n_planes = (uint32_t)L->planes.size();
    d.boxes = L->boxes.data(); d.n_boxes = (uint32_t)L->boxes.size();
    d.meshes = L->meshes.data(); d.n_meshes = (uint32_t)L->meshes.size();
    d.lights = L->lights.data(); d.n_lights = (uint32_t)L->lights.size();
    return L.release();
}

void free_desc(LoadedDesc* l) { delete l; }
const rt_scene_desc* desc_of(LoadedDesc* l) { return &l->desc; }

bool Scene::save(const std::string& path, std::string& err) {
    const rt_scene_desc* d = flatten(err);
    if (!d) return false;
    return save_desc(d, path, err);
}

}  // namespace rth
