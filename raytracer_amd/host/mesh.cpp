// Host-side mesh construction: the steps the reference runs once per mesh
// before the render path (object/mesh.rs:135-324, acceleration.rs:175-448).
// They are restated so that the structure the device traverses -- the top-down
// binned-SAH tree -- has the same topology, leaf contents and boxes the
// reference's build produces for the same triangles.
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <sstream>
#include <unordered_map>

#include "scene.hpp"

namespace rth {

// ---------------------------------------------------------------- transforms
void Mesh::shift(Vector3 delta) { for (auto& v : verts) { v.x += delta.x; v.y += delta.y; v.z += delta.z; } }   // mesh.rs:275-277
void Mesh::scale(double delta) { for (auto& v : verts) { v.x *= delta; v.y *= delta; v.z *= delta; } }          // mesh.rs:296-298

void Mesh::center() {   // mesh.rs:279-293
    const double MAXF = 1.7976931348623157e308;
    Vector3 mn(MAXF, MAXF, MAXF), mx(-MAXF, -MAXF, -MAXF);
    for (const auto& v : verts) {
        mn.x = std::fmin(mn.x, v.x); mn.y = std::fmin(mn.y, v.y); mn.z = std::fmin(mn.z, v.z);
        mx.x = std::fmax(mx.x, v.x); mx.y = std::fmax(mx.y, v.y); mx.z = std::fmax(mx.z, v.z);
    }
    shift((mn + mx) * -0.5);
}

static void rotate_with(Mesh& m, const Matrix& rot) {
    for (auto& v : m.verts) v = (rot * Matrix::from_pos(v)).pos();
    for (auto& n : m.normals) n = (rot * Matrix::from_pos(n)).pos();
}
void Mesh::rotate_xyz(Vector3 r) { rotate_with(*this, from_euler_xyz(-r.x, -r.y, -r.z)); }   // mesh.rs:301-311
void Mesh::rotate_zyx(Vector3 r) { rotate_with(*this, from_euler_zyx(-r.x, -r.y, -r.z)); }   // mesh.rs:314-324

static Vector3 triangle_normal(Vector3 v0, Vector3 v1, Vector3 v2) {   // mesh.rs:19-23
    return normalize(cross(v1 - v0, v2 - v0));
}

// mesh.rs:208-254.  The reference walks a HashMap, so the ORDER of the emitted
// normals is unspecified there; the value attached to each triangle corner is
// not.  Vertices are visited in ascending index order here.
void Mesh::recalculate_normals() {
    normals.clear();
    tri_normals.assign(tris.size(), {0u, 0u, 0u});
    std::vector<std::vector<uint32_t>> vert_map(verts.size());
    for (uint32_t idx = 0; idx < tris.size(); idx++)
        for (uint32_t c : tris[idx])
            if (c < vert_map.size()) vert_map[c].push_back(idx);
    for (uint32_t vert_idx = 0; vert_idx < vert_map.size(); vert_idx++) {
        const auto& adj = vert_map[vert_idx];
        if (adj.empty()) continue;
        Vector3 agg;
        for (uint32_t t : adj) {
            const auto& tri = tris[t];
            Vector3 n = triangle_normal(verts[tri[0]], verts[tri[1]], verts[tri[2]]);
            agg.x += n.x; agg.y += n.y; agg.z += n.z;
        }
        normals.push_back(normalize(agg / (double)adj.size()));
        for (uint32_t t : adj)
            for (int n = 0; n < 3; n++)
                if (tris[t][n] == vert_idx) tri_normals[t][n] = (uint32_t)normals.size() - 1;
    }
}

// ---------------------------------------------------------------- BVH build
namespace {
constexpr int OBJECT_BUCKETS = 32;   // acceleration.rs:20
constexpr double FMAX = 1.7976931348623157e308;

struct Box {   // acceleration.rs:66-71, Default :154-162
    Vector3 centroid{0., 0., 0.}, min{FMAX, FMAX, FMAX}, max{-FMAX, -FMAX, -FMAX};
    Box unite(const Box& o) const {   // :103-120
        Box r;
        r.min = Vector3(std::fmin(min.x, o.min.x), std::fmin(min.y, o.min.y), std::fmin(min.z, o.min.z));
        r.max = Vector3(std::fmax(max.x, o.max.x), std::fmax(max.y, o.max.y), std::fmax(max.z, o.max.z));
        r.centroid = (r.min + r.max) * 0.5;
        return r;
    }
    double surface_area() const {     // :122-128
        double xs = max.x - min.x, ys = max.y - min.y, zs = max.z - min.z;
        return 2. * xs * ys + 2. * xs * zs + 2. * ys * zs;
    }
    double extent(int a) const { return max.axis(a) - min.axis(a); }   // :130-132
};

// Rust `as usize` from f64: saturating, NaN -> 0
inline uint64_t as_usize(double v) { if (!(v > 0.0)) return 0; if (v >= 18446744073709551615.0) return UINT64_MAX; return (uint64_t)v; }

struct Builder {
    const std::vector<Box>& refs;
    std::string err;
    std::atomic<bool> failed{false};
    explicit Builder(const std::vector<Box>& r) : refs(r) {}

    inline uint64_t bucket_of(uint32_t idx, const Box& total, int axis, double extent) const {
        return as_usize((refs[idx].centroid.axis(axis) - total.min.axis(axis)) / extent * (double)OBJECT_BUCKETS);
    }

    // acceleration.rs:199-253
    bool split_candidate(const Box& total, const uint32_t* input, size_t n, int axis, double& cost_out, int& idx_out) {
        struct Bucket { Box box; size_t count = 0; } buckets[OBJECT_BUCKETS];
        const double extent = total.extent(axis);
        for (size_t k = 0; k < n; k++) {
            uint64_t b = bucket_of(input[k], total, axis, extent);
            if (b >= (uint64_t)OBJECT_BUCKETS) return false;   // the reference indexes out of bounds here and panics
            buckets[b].count += 1;
            buckets[b].box = buckets[b].box.unite(refs[input[k]]);
        }
        struct Side { size_t n = 0; Box box; } lhs_splits[OBJECT_BUCKETS - 1], lhs, rhs;
        for (int i = 0; i < OBJECT_BUCKETS - 1; i++) {
            lhs.n += buckets[i].count;
            lhs.box = lhs.box.unite(buckets[i].box);
            lhs_splits[i] = lhs;
        }
        double min_cost = FMAX;
        int min_idx = 0;
        for (int i = OBJECT_BUCKETS - 1; i >= 1; i--) {
            rhs.n += buckets[i].count;
            rhs.box = rhs.box.unite(buckets[i].box);
            const Side& l = lhs_splits[i - 1];
            const double traverse_cost = 1.;
            double n_lhs = (double)l.n, n_rhs = (double)rhs.n;
            double cost = traverse_cost + (n_lhs * l.box.surface_area() + n_rhs * rhs.box.surface_area()) / total.surface_area();
            if (cost < min_cost) { min_cost = cost; min_idx = i; }
        }
        cost_out = min_cost; idx_out = min_idx;
        return true;
    }
};

struct TreeN { std::unique_ptr<TreeN> a, b; std::vector<uint32_t> refs; };

// acceleration.rs:323-378; children of large ranges are built in parallel
// (the reference uses rayon::join, :369-372)
std::unique_ptr<TreeN> build_rec(Builder& B, uint32_t* input, uint32_t* output, size_t n, int par_depth) {
    if (B.failed.load(std::memory_order_relaxed)) return nullptr;
    auto leaf = [&]() {   // acceleration.rs:380-384
        auto t = std::make_unique<TreeN>();
        t->refs.assign(input, input + n);
        return t;
    };
    if (n < 2) return leaf();   // :329-331
    Box total;
    for (size_t k = 0; k < n; k++) total = total.unite(B.refs[input[k]]);   // :333-336
    double ex = total.extent(0), ey = total.extent(1), ez = total.extent(2);
    int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);             // :338-349
    double leaf_cost = (double)n;
    double cost; int bucket;
    if (!B.split_candidate(total, input, n, axis, cost, bucket)) {
        B.failed.store(true);
        return nullptr;
    }
    if (cost < leaf_cost) {   // :356
        // object_split, :255-283: left part fills from the front in input order,
        // right part from the back (so it ends up reversed)
        const double extent = total.extent(axis);
        size_t lhs_ptr = 0, rhs_ptr = n - 1;
        for (size_t k = 0; k < n; k++) {
            uint64_t b = B.bucket_of(input[k], total, axis, extent);
            if (b < (uint64_t)bucket) output[lhs_ptr++] = input[k];
            else output[rhs_ptr--] = input[k];
        }
        size_t split = lhs_ptr;
        auto t = std::make_unique<TreeN>();
        // children read `output` and use `input` as their scratch (:366-367)
        if (par_depth > 0 && n > 20000) {
            auto fa = std::async(std::launch::async, [&]() { return build_rec(B, output, input, split, par_depth - 1); });
            t->b = build_rec(B, output + split, input + split, n - split, par_depth - 1);
            t->a = fa.get();
        } else {
            t->a = build_rec(B, output, input, split, 0);
            t->b = build_rec(B, output + split, input + split, n - split, 0);
        }
        return t;
    }
    return leaf();   // :375-376
}

// acceleration.rs:386-400,419-433: every node's box is the union of the boxes
// of the leaf references below it; nodes are emitted depth-first, a before b.
Box flatten_rec(const TreeN& t, const std::vector<Box>& refs, FlatBvh& out, uint32_t depth) {
    if (depth > out.depth) out.depth = depth;
    uint32_t me = (uint32_t)out.nodes.size();
    out.nodes.push_back(rt_bvh_node{});
    Box box;
    if (!t.a) {
        uint32_t first = (uint32_t)out.prim_indices.size();
        for (uint32_t r : t.refs) { box = box.unite(refs[r]); out.prim_indices.push_back(r); }
        out.nodes[me].right_or_first = first;
        out.nodes[me].count = (uint32_t)t.refs.size();
    } else {
        Box la = flatten_rec(*t.a, refs, out, depth + 1);
        uint32_t right = (uint32_t)out.nodes.size();
        Box lb = flatten_rec(*t.b, refs, out, depth + 1);
        box = la.unite(lb);
        out.nodes[me].right_or_first = right;
        out.nodes[me].count = 0;
    }
    rt_bvh_node& nd = out.nodes[me];
    nd.min[0] = box.min.x; nd.min[1] = box.min.y; nd.min[2] = box.min.z;
    nd.max[0] = box.max.x; nd.max[1] = box.max.y; nd.max[2] = box.max.z;
    return box;
}
}  // namespace

// mesh.rs:257-272 + acceleration.rs:296-321, 435-439
bool Mesh::generate_sbvh(std::string& err) {
    sbvh.reset();
    const size_t n = tris.size();
    if (n == 0) { err = "generate_sbvh: mesh has no triangles (the reference underflows 2*0-1, acceleration.rs:313)"; return false; }
    std::vector<Box> refs(n);
    for (size_t i = 0; i < n; i++) {   // acceleration.rs:457-463, Aabb::from_vecs :82-101
        Box b;
        for (int c = 0; c < 3; c++) {
            uint32_t vi = tris[i][c];
            if (vi >= verts.size()) { err = "generate_sbvh: vertex index out of range"; return false; }
            const Vector3& v = verts[vi];
            b.min = Vector3(std::fmin(b.min.x, v.x), std::fmin(b.min.y, v.y), std::fmin(b.min.z, v.z));
            b.max = Vector3(std::fmax(b.max.x, v.x), std::fmax(b.max.y, v.y), std::fmax(b.max.z, v.z));
        }
        b.centroid = (b.min + b.max) * 0.5;
        refs[i] = b;
    }
    std::vector<uint32_t> buffer_a(n), buffer_b(n, 0);
    for (size_t i = 0; i < n; i++) buffer_a[i] = (uint32_t)i;
    Builder B(refs);
    auto root = build_rec(B, buffer_a.data(), buffer_b.data(), n, 3);
    if (B.failed.load() || !root) {
        err = "generate_sbvh: a primitive centroid lies on the far face of its node on the split axis; "
              "the reference's bucket index reaches 32 and it panics (acceleration.rs:210-215)";
        return false;
    }
    auto flat = std::make_shared<FlatBvh>();
    flat->nodes.reserve(2 * n);
    flat->prim_indices.reserve(n);
    flatten_rec(*root, refs, *flat, 1);
    sbvh = flat;
    return true;
}

// ---------------------------------------------------------------- OBJ import
// mesh.rs:135-205 on top of tobj 3.2.0's load_obj(triangulate = true), first
// model only: f32 attribute parsing, 1-based and negative (relative) indices,
// fans for polygons, positions / texcoords / normals re-indexed separately in
// order of first use (tobj's multi-index export).
namespace {
struct VRef { int64_t v = -1, vt = -1, vn = -1; };

bool parse_vref(const char* tok, size_t nv, size_t nvt, size_t nvn, VRef& out) {
    long idx[3] = {0, 0, 0};
    bool has[3] = {false, false, false};
    int field = 0;
    const char* p = tok;
    while (*p && field < 3) {
        if (*p == '/') { field++; p++; continue; }
        char* end = nullptr;
        long val = std::strtol(p, &end, 10);
        if (end == p) return false;
        idx[field] = val; has[field] = true;
        p = end;
    }
    size_t counts[3] = {nv, nvt, nvn};
    int64_t res[3] = {-1, -1, -1};
    for (int f = 0; f < 3; f++) {
        if (!has[f]) continue;
        if (idx[f] > 0) res[f] = idx[f] - 1;
        else if (idx[f] < 0) res[f] = (int64_t)counts[f] + idx[f];
        else return false;
        if (res[f] < 0) return false;
    }
    if (!has[0]) return false;
    out.v = res[0]; out.vt = res[1]; out.vn = res[2];
    return true;
}
}  // namespace

bool Mesh::from_obj(const std::string& file_name, Material material, Mesh& out, std::string& err) {
    std::ifstream in(file_name);
    if (!in) { err = "failed to parse obj: cannot open " + file_name; return false; }
    std::vector<float> pos, tex, nrm;
    std::vector<std::vector<VRef>> faces;
    std::string line;
    bool model_closed = false;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag[0] == '#') continue;
        if (tag == "v") {
            std::string a, b, c; ss >> a >> b >> c;
            pos.push_back(std::strtof(a.c_str(), nullptr)); pos.push_back(std::strtof(b.c_str(), nullptr)); pos.push_back(std::strtof(c.c_str(), nullptr));
        } else if (tag == "vt") {
            std::string a, b; ss >> a >> b;
            tex.push_back(std::strtof(a.c_str(), nullptr)); tex.push_back(b.empty() ? 0.f : std::strtof(b.c_str(), nullptr));
        } else if (tag == "vn") {
            std::string a, b, c; ss >> a >> b >> c;
            nrm.push_back(std::strtof(a.c_str(), nullptr)); nrm.push_back(std::strtof(b.c_str(), nullptr)); nrm.push_back(std::strtof(c.c_str(), nullptr));
        } else if (tag == "f") {
            if (model_closed) continue;
            std::vector<VRef> f;
            std::string tok;
            while (ss >> tok) {
                VRef r;
                if (!parse_vref(tok.c_str(), pos.size() / 3, tex.size() / 2, nrm.size() / 3, r)) { err = "failed to parse obj: bad face element '" + tok + "'"; return false; }
                f.push_back(r);
            }
            faces.push_back(std::move(f));
        } else if (tag == "o" || tag == "g") {
            // a new object/group after faces ends the first model (the only one the reference keeps, mesh.rs:145)
            if (!faces.empty()) model_closed = true;
        }
    }
    if (faces.empty()) { err = "failed to parse obj: no faces in " + file_name; return false; }

    Mesh m;
    m.material = std::move(material);
    std::unordered_map<int64_t, uint32_t> vmap, tmap, nmap;
    auto add = [&](const VRef& r, std::array<uint32_t, 3>& tp, std::array<uint32_t, 3>& tt, std::array<uint32_t, 3>& tn, int k, bool& has_t, bool& has_n) -> bool {
        if ((size_t)r.v * 3 + 2 >= pos.size()) return false;
        auto it = vmap.find(r.v);
        if (it == vmap.end()) {
            uint32_t id = (uint32_t)m.verts.size();
            m.verts.emplace_back((double)pos[3 * r.v], (double)pos[3 * r.v + 1], (double)pos[3 * r.v + 2]);   // mesh.rs:152
            vmap.emplace(r.v, id); tp[k] = id;
        } else tp[k] = it->second;
        if (r.vt >= 0 && !tex.empty()) {
            if ((size_t)r.vt * 2 + 1 >= tex.size()) return false;
            auto jt = tmap.find(r.vt);
            if (jt == tmap.end()) {
                uint32_t id = (uint32_t)m.texcoords.size();
                m.texcoords.push_back({tex[2 * r.vt], tex[2 * r.vt + 1]});
                tmap.emplace(r.vt, id); tt[k] = id;
            } else tt[k] = jt->second;
            has_t = true;
        }
        if (r.vn >= 0 && !nrm.empty()) {
            if ((size_t)r.vn * 3 + 2 >= nrm.size()) return false;
            auto jt = nmap.find(r.vn);
            if (jt == nmap.end()) {
                uint32_t id = (uint32_t)m.normals.size();
                m.normals.emplace_back((double)nrm[3 * r.vn], (double)nrm[3 * r.vn + 1], (double)nrm[3 * r.vn + 2]);   // mesh.rs:168
                nmap.emplace(r.vn, id); tn[k] = id;
            } else tn[k] = jt->second;
            has_n = true;
        }
        return true;
    };
    for (const auto& f : faces) {
        if (f.size() < 3) continue;   // points and lines carry no triangles
        for (size_t c = 2; c < f.size(); c++) {   // fan: (0, c-1, c)
            std::array<uint32_t, 3> tp{0, 0, 0}, tt{0, 0, 0}, tn{0, 0, 0};
            bool has_t = false, has_n = false;
            const VRef* vr[3] = {&f[0], &f[c - 1], &f[c]};
            for (int k = 0; k < 3; k++)
                if (!add(*vr[k], tp, tt, tn, k, has_t, has_n)) { err = "failed to parse obj: face vertex out of bounds"; return false; }
            m.tris.push_back(tp);
            if (has_t) m.tri_texcoords.push_back(tt);
            if (has_n) m.tri_normals.push_back(tn);
        }
    }
    if (!m.tri_texcoords.empty() && m.tri_texcoords.size() != m.tris.size()) { err = "obj: faces mix vertices with and without texture coordinates"; return false; }
    if (!m.tri_normals.empty() && m.tri_normals.size() != m.tris.size()) { err = "obj: faces mix vertices with and without normals"; return false; }
    out = std::move(m);
    return true;
}

}  // namespace rth
