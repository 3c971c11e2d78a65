// librtgpu.so: the C ABI of include/rtgpu.h on HIP for gfx950.
//
// Host side of the render path: scene upload (deep copy + re-emission of each
// mesh BVH in the device layout), the per-frame wavefront loop over trace_ray
// depths, and the copy-out.  There is no CPU rendering path in this library: a
// missing device is an error.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is opened when the first context over several GPUs is created
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/rtgpu.h"
#include "kernels.h"
#include "bvh_build.h"

namespace {

thread_local std::string g_create_error = "";

struct LevelAlloc { void* base = nullptr; size_t cap = 0; bool keys = false; DLevel lv; };

// Per-level buffers and events of the shadow side of a level, which runs beside the trace of the next level
struct LevelWork {
    void* hitlist = nullptr; size_t hitlist_cap = 0;   // rays of the level whose hit is lit
    void* occ = nullptr; size_t occ_cap = 0;           // shadow results [light][ray]
    void* hs = nullptr; size_t hs_cap = 0;             // hit point + shadow-ray origin of every lit hit, 6 planes of doubles
    void* long_trace = nullptr; size_t long_trace_cap = 0;     // long-ray lists
    void* long_shadow = nullptr; size_t long_shadow_cap = 0;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // see LW_*
};
enum { LW_T0, LW_T1, LW_P, LW_S0, LW_SK, LW_S1, LW_L0, LW_L1 };
#define RT_MAX_LEVELS 33  /* trace_ray depths 0..max_ray_depth (at most 31; the reference's own scenes go up to 8, scenes/fedora_light.sdl:6), and one more for the counters' sake */
#define RT_SIDE_STREAMS 4

}  // namespace

// ---- rt_cast_rays_noalloc: staging for the probe rays and their hit records in a file-scope device array instead of
// per-call hipMalloc.  Every GPU the library runs on holds its own copy of the array (at its own address) until the
// process ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress -- it is not a device
// address -- so the address is looked up once per device (rt_ctx_create, with that device current) and everything
// below works with the looked-up pointer.
#define RT_NOALLOC_ORIGINS_OFF 0u
#define RT_NOALLOC_DIRS_OFF    ((size_t)RT_CAST_NOALLOC_MAX * 24u)
#define RT_NOALLOC_HITS_OFF    ((size_t)RT_CAST_NOALLOC_MAX * 48u)
#define RT_NOALLOC_BYTES       ((size_t)RT_CAST_NOALLOC_MAX * (48u + sizeof(rt_ray_hit)))
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
namespace {
#define RT_MAX_DEVICES 64
std::mutex g_scratch_mutex;
void* g_scratch_ptr[RT_MAX_DEVICES] = {};   // device address of `scratch` on each device, once looked up

// the calling thread's current device must be `device`
hipError_t scratch_address(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct rt_ctx {
    int device = 0;
    void* scratch = nullptr;       // this device's copy of the file-scope staging array (rt_cast_rays_noalloc)
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> scene_allocs;
    std::vector<void*>* alloc_sink = &scene_allocs;   // where upload() records what it allocates
    // the meshes of the last upload, each with the device memory it owns: a mesh flagged `unchanged` keeps its copy
    struct MeshCopy {
        DMesh dm; std::vector<void*> allocs; size_t bytes = 0; uint32_t n_verts = 0, n_tris = 0, n_nodes = 0, n_prims = 0;
        uint32_t n_normals = 0;
        uint32_t n_records = 0;                // triangle records on the device: one per leaf entry of a caller's tree (n_prims), one per triangle of a device-built one
        const uint32_t* d_tri_pos = nullptr;   // the triangles' vertex indices, kept for refits (rt_mesh::unchanged == 2)
    };
    std::vector<MeshCopy> mesh_copies;
    DScene hscene;                 // host copy of the device scene header
    DScene* dscene = nullptr;
    bool have_scene = false;
    int32_t vw = 0, vh = 0;
    uint32_t max_depth = 0;
    std::vector<LevelAlloc> levels;
    DCounters* dcounters = nullptr;   // [RT_MAX_LEVELS], one set per trace_ray depth
    DCounters* hcounters = nullptr;   // pinned
    void* d_out = nullptr; size_t d_out_cap = 0;      // rgb staging for rt_render
    void* h_stage = nullptr; size_t h_stage_cap = 0;  // pinned host staging for rt_render's copy of the frame (copy_frame_to_host)
    unsigned int* h_flags = nullptr;                  // pinned: one word per block of that copy, raised by the device when the block has landed
    uint32_t copy_threads = 4;                        // RTGPU_COPY_THREADS: host threads that move the chunks on (0: one plain hipMemcpy)
    void* d_ids = nullptr; size_t d_ids_cap = 0;      // compact ids
    void* d_ids2 = nullptr; size_t d_ids2_cap = 0;    // row-major ids
    void* d_cams = nullptr; size_t d_cams_cap = 0;    // cameras of a batch of frames (rt_render_shard_batch)
    std::vector<DCamera> h_cams;
    std::vector<LevelWork> lw;
    DLevelCounts* d_lcounts = nullptr;                // [n_lcounts], one pair of cursors per level
    DLevelCounts* h_lcounts = nullptr;                // pinned
    size_t n_lcounts = 0;
    // Streams in use (RTGPU_SIDE_STREAMS lowers it).  A context has 1 + n_side streams; when the streams of all the contexts
    // of a process outnumber its hardware queues (GPU_MAX_HW_QUEUES), some share a queue and run one after the other: a
    // pipeline of 4 contexts that shares unluckily is 15 % slower.  4 contexts x 5 streams want 24 queues.
    uint32_t n_side = RT_SIDE_STREAMS;
    hipStream_t side[RT_SIDE_STREAMS] = {nullptr, nullptr, nullptr, nullptr};   // shadow + lighting of level d runs on side[d % N]
    LongParams long_params = {0u, 256u};              // walk budgets (inner nodes per lane / per wave) before a ray goes on the long-ray list; 0 = off
    bool any_need_far = false;                        // some mesh's material needs Hit::far (transparency with ior == 1)
    bool rays_need_reciprocal = true;              // the packets meet a cube or the object tree: their rays carry 1/direction (k_shadow, k_trace_deferred <.., RECIP>)
    bool any_mesh_refracts = false;                   // some mesh has a refracting material (transparency with ior != 1): k_spawn needs the mesh walk
    uint32_t packet_depth = 0;                        // levels 0..packet_depth trace their rays as wave packets
    uint32_t defer_objects_below = 0;                 // scenes without meshes: levels below this one walk the object tree as deferred packets (the others lane by lane)
    bool defer_from_set = false;                      // RTGPU_DEFER_FROM given: no rule of thumb for level 0
    uint32_t passes = 0;                              // RTGPU_PASSES: passes over a whole frame (0: by its size, see render_frame)
    uint64_t total_tris = 0;                          // triangles of all meshes of the uploaded scene
    bool defer_small = true;                          // ... also in launches small enough for the long-ray budgets
    uint32_t defer_from = 1;                          // levels from this one on walk meshes as packets with the triangle tests and the thin ends set aside (scenes without need_far meshes)
    bool serial = false;                              // RTGPU_SERIAL=1: everything on the main stream (profiling: kernels do not overlap)
    uint32_t long_below = 300000u;                  // ... applied to launches of fewer rays than this
    int count_visits = 0;                             // RTGPU_COUNT_VISITS: bit 0 k_trace, bit 1 k_shadow, bit 2 k_shade instrumented
    uint32_t n_lights = 0;                            // light records on the device: one per sun and point light, one per sample of an area light
    bool has_area = false;                            // the scene has area lights
    uint64_t area_seed = 0x9E3779B97F4A7C15ull;       // seed of their sample points (rt_set_area_seed)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // level sizes of the last frame rendered for (shard, n_shards, vw, vh, max_depth): what the next such frame is given room for
    typedef std::tuple<uint32_t, uint32_t, int32_t, int32_t, uint32_t, uint32_t> LearnedKey;
    std::map<LearnedKey, std::vector<uint64_t>> learned;
    // rays of all levels of one whole frame, as last rendered, for (vw, vh, max_depth): what the choice of passes over a
    // whole frame goes by (render_frame)
    std::map<std::tuple<int32_t, int32_t, uint32_t>, uint64_t> whole_frame_rays;
    uint64_t scene_shape = 0;                         // counts of the last uploaded scene (objects, lights, meshes, triangles): see scene_upload
    bool sync_levels = false;                         // RTGPU_SYNC_LEVELS=1: always read each level's size back before the next
    uint64_t respeculated = 0;                        // frames rendered twice because a level outgrew its room
    size_t scene_bytes = 0, work_bytes = 0;
    double ms_upload = 0.;
    struct rt_multi* multi = nullptr;                 // a context over several GPUs (rt_ctx_create_multi): this one is its first member
    struct rt_multi* solo = nullptr;                  // a one-device context's buffers for sets of frames (rt_render_frames), made on first use
};

namespace {

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return RT_ERR_DEVICE;                                                                   \
        }                                                                                           \
    } while (0)

int fail(rt_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }

void free_mesh_copies(std::vector<rt_ctx::MeshCopy>& v) {
    for (auto& m : v) for (void* p : m.allocs) (void)hipFree(p);
    v.clear();
}
void free_scene(rt_ctx* c) {
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->scene_bytes = 0;
    c->have_scene = false;
}

template <typename T>
int upload(rt_ctx* c, const T* src, size_t n, const T** out) {
    *out = nullptr;
    if (n == 0) return RT_OK;
    void* d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc failed during scene upload");
    c->alloc_sink->push_back(d);
    c->scene_bytes += n * sizeof(T);
    HIP_TRY(c, hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return RT_OK;
}

uint32_t pack_rgb(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

// matrix.rs:148-185 (4x4 product accumulated from 0.0 in k order) for the
// camera rotation the reference rebuilds per pixel (camera.rs:56-57)
struct M4 { double a[4][4]; };
M4 m4mul(const M4& A, const M4& B) {
    M4 o;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            o.a[i][j] = 0.;
            for (int k = 0; k < 4; k++) o.a[i][j] += A.a[i][k] * B.a[k][j];
        }
    // the reference rebuilds a Matrix from o, dropping row 3 (:170-183); row 3 of the next operand is again 0,0,0,1
    o.a[3][0] = 0.; o.a[3][1] = 0.; o.a[3][2] = 0.; o.a[3][3] = 1.;
    return o;
}
void camera_rotation(double pitch, double yaw, double rot[3][3]) {
    // from_euler_xyz(-pitch, yaw, 0) = a * b * c, matrix.rs:60-72
    // Sine and cosine of an angle from ONE sincos() call: compilers merge a sin/cos pair of the same argument into it
    // (or not), glibc's sincos and sin can differ in the last bit, and a ray direction that is one ulp off shows up as
    // a grey level where a blend weight of exactly 1 lands the colour on an integer.
    const double x = -pitch, y = yaw, z = 0.;
    double sx, cx, sy, cy, sz, cz;
    ::sincos(x, &sx, &cx); ::sincos(y, &sy, &cy); ::sincos(z, &sz, &cz);
    M4 a = {{{1., 0., 0., 0.}, {0., cx, -sx, 0.}, {0., sx, cx, 0.}, {0., 0., 0., 1.}}};
    M4 b = {{{cy, 0., sy, 0.}, {0., 1., 0., 0.}, {-sy, 0., cy, 0.}, {0., 0., 0., 1.}}};
    M4 cc = {{{cz, -sz, 0., 0.}, {sz, cz, 0., 0.}, {0., 0., 1., 0.}, {0., 0., 0., 1.}}};
    M4 r = m4mul(m4mul(a, b), cc);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rot[i][j] = r.a[i][j];
}

DCamera make_camera(const rt_camera& cam) {
    DCamera d;
    for (int k = 0; k < 3; k++) d.cam_o[k] = cam.origin[k];
    camera_rotation(cam.pitch, cam.yaw, d.rot);
    d.half_w = (double)cam.vw * 0.5;               // camera.rs:63
    d.half_h = (double)cam.vh * 0.5;               // :64
    d.cam_z = (double)cam.vh * 0.5 * cam.chf;      // :65
    return d;
}

// Re-emits one mesh: DFS-preorder host nodes (rt_bvh_node) -> inner-node array
// with both children's boxes, triangles in leaf order with precomputed edges.
// fn(begin, end) over [0, n) on a few host threads (scene upload: re-emitting a million-triangle tree is host work)
template <typename F>
void parallel_for(size_t n, F fn) {
    size_t nt = std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), 16, n / 16384 + 1});
    if (const char* e = std::getenv("RTGPU_UPLOAD_THREADS")) nt = std::max<size_t>(1, std::strtoul(e, nullptr, 10));
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; t++) {
        const size_t a = std::min(n, t * per), b = std::min(n, a + per);
        if (a < b) th.emplace_back([=]() { fn(a, b); });
    }
    for (auto& t : th) t.join();
}

// f32 root box and coordinate bound of a mesh from its f64 bounds: rounded outward from the slab bounds the reference's
// test uses, centroid -+ size with centroid = (min + max) * 0.5 and size = max - centroid (acceleration.rs:76,135)
void set_root_box(DMesh& out, const double mn[3], const double mx[3]) {
    for (int k = 0; k < 3; k++) {
        const double cen = (mn[k] + mx[k]) * 0.5;
        const double half = mx[k] - cen;
        const double lo = std::fmin(mn[k], cen - half), hi = std::fmax(mx[k], cen + half);
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
        if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
        out.root_min[k] = flo; out.root_max[k] = fhi;
    }
    // every child box lies inside the root's (boxes are unions of their descendants'), up to the outward rounding
    out.coord_max = 0.f;
    for (int k = 0; k < 3; k++) out.coord_max = std::fmax(out.coord_max, std::fmax(std::fabs(out.root_min[k]), std::fabs(out.root_max[k])));
    out.coord_max = out.coord_max * 1.000001f + 1e-30f;
}

int check_mesh_indices(rt_ctx* c, const rt_mesh& m) {
    std::atomic<int> bad{0};   // 1 vertex, 2 normal, 3 texcoord
    parallel_for((size_t)3 * m.n_tris, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            if (m.tri_pos[i] >= m.n_verts) { bad = 1; return; }
            if (m.tri_nrm[i] >= m.n_normals) { bad = 2; return; }
            if (m.tri_uv && m.n_texcoords && m.tri_uv[i] >= m.n_texcoords) { bad = 3; return; }
        }
    });
    if (bad == 1) return fail(c, RT_ERR_INVALID, "mesh: vertex index out of range");
    if (bad == 2) return fail(c, RT_ERR_INVALID, "mesh: normal index out of range");
    if (bad == 3) return fail(c, RT_ERR_INVALID, "mesh: texcoord index out of range");
    return RT_OK;
}

// the per-triangle attribute arrays every mesh carries, whoever built its tree
int upload_mesh_attributes(rt_ctx* c, const rt_mesh& m, rt_ctx::MeshCopy& mc) {
    DMesh& out = mc.dm;
    int rc;
    out.n_tris = m.n_tris; out.n_texcoords = m.tri_uv ? m.n_texcoords : 0;
    if ((rc = upload(c, m.normals, (size_t)m.n_normals * 3, &out.normals)) != RT_OK) return rc;
    if ((rc = upload(c, m.texcoords, (size_t)m.n_texcoords * 2, &out.texcoords)) != RT_OK) return rc;
    if ((rc = upload(c, m.tri_nrm, (size_t)m.n_tris * 3, &out.tri_nrm)) != RT_OK) return rc;
    if (m.tri_uv && m.n_texcoords) { if ((rc = upload(c, m.tri_uv, (size_t)m.n_tris * 3, &out.tri_uv)) != RT_OK) return rc; }
    if (!mc.d_tri_pos && (rc = upload(c, m.tri_pos, (size_t)m.n_tris * 3, &mc.d_tri_pos)) != RT_OK) return rc;
    mc.n_normals = m.n_normals;
    return RT_OK;
}

// rt_mesh::nodes == NULL: the tree is built here, on the device (bvh_build.h)
int build_mesh_on_device(rt_ctx* c, const rt_mesh& m, rt_ctx::MeshCopy& mc) {
    DMesh& out = mc.dm;
    std::memset(&out, 0, sizeof(out));
    if (m.n_tris == 0 || !m.tri_pos || !m.tri_nrm || !m.verts) return fail(c, RT_ERR_INVALID, "mesh without triangles");
    if (m.n_tris > RT_PACKET_MAX_PRIMS) return fail(c, RT_ERR_UNSUPPORTED, "mesh with more than " + std::to_string(RT_PACKET_MAX_PRIMS) + " triangles");
    if (m.n_normals != 0 && !m.normals) return fail(c, RT_ERR_INVALID, "mesh: normals is NULL but n_normals is not 0");
    if (m.n_texcoords != 0 && !m.texcoords) return fail(c, RT_ERR_INVALID, "mesh: texcoords is NULL but n_texcoords is not 0");
    int rc;
    if ((rc = check_mesh_indices(c, m)) != RT_OK) return rc;
    if ((rc = upload(c, m.tri_pos, (size_t)m.n_tris * 3, &mc.d_tri_pos)) != RT_OK) return rc;
    double* d_verts = nullptr;
    if (hipMalloc((void**)&d_verts, (size_t)m.n_verts * 24) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc failed during scene upload");
    hipError_t e = hipMemcpyAsync(d_verts, m.verts, (size_t)m.n_verts * 24, hipMemcpyHostToDevice, c->stream);
    BvhBuildResult br;
    if (e == hipSuccess) e = bvh_build_device(c->stream, d_verts, mc.d_tri_pos, m.n_tris, &br);
    (void)hipFree(d_verts);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, std::string("device BVH build: ") + hipGetErrorString(e));
    if (br.nodes) { c->alloc_sink->push_back(br.nodes); c->scene_bytes += (size_t)br.n_inner * sizeof(DNode); }
    c->alloc_sink->push_back(br.tris); c->scene_bytes += (size_t)m.n_tris * sizeof(DTri);
    if (std::getenv("RTGPU_DEBUG_BVH")) std::fprintf(stderr, "device-built tree: wide nodes %u, stack %u, sort %.2f ms, total %.2f ms\n", br.n_inner, br.stack_need, br.ms_sort, br.ms_total);
    if (br.stack_need > RT_PSTACK)
        return fail(c, RT_ERR_UNSUPPORTED, "device-built BVH needs " + std::to_string(br.stack_need) + " stack entries (limit " + std::to_string(RT_PSTACK) +
                                               "): build this mesh's tree on the host");
    out.nodes = br.nodes; out.tris = br.tris; out.n_inner = br.n_inner; out.root_ref = br.root_ref;
    mc.n_records = m.n_tris;
    set_root_box(out, br.root_box, br.root_box + 3);
    return upload_mesh_attributes(c, m, mc);
}

// rt_mesh::unchanged == 2: same connectivity and tree as the kept copy, new vertex positions (and normals): triangle
// records and boxes are recomputed on the device
int refit_mesh(rt_ctx* c, const rt_mesh& m, rt_ctx::MeshCopy& mc) {
    if (!m.verts) return fail(c, RT_ERR_INVALID, "mesh to refit without vertices");
    if (m.n_normals != mc.n_normals || (m.n_normals && !m.normals)) return fail(c, RT_ERR_INVALID, "mesh to refit: normals differ in number from the kept copy's");
    if (mc.n_records == 0) return fail(c, RT_ERR_INVALID, "mesh to refit: the kept copy holds no triangle records");
    double* d_verts = nullptr;
    if (hipMalloc((void**)&d_verts, (size_t)m.n_verts * 24) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc failed during scene upload");
    hipError_t e = hipMemcpyAsync(d_verts, m.verts, (size_t)m.n_verts * 24, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && m.n_normals)
        e = hipMemcpyAsync(const_cast<double*>(mc.dm.normals), m.normals, (size_t)m.n_normals * 24, hipMemcpyHostToDevice, c->stream);
    double root_box[6];
    if (e == hipSuccess)
        e = bvh_refit_device(c->stream, d_verts, mc.d_tri_pos, mc.n_records, const_cast<DNode*>(mc.dm.nodes), mc.dm.n_inner, mc.dm.root_ref,
                             const_cast<DTri*>(mc.dm.tris), root_box);
    (void)hipFree(d_verts);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, std::string("device BVH refit: ") + hipGetErrorString(e));
    set_root_box(mc.dm, root_box, root_box + 3);
    return RT_OK;
}

// A caller's tree that is too deep for the traversal stacks is not walked as it stands: its leaves, in the tree's own
// depth-first order, get a new binary tree above them, split by leaf count (depth log2 of the leaf count), each node's
// box the union of its leaves' boxes -- what the reference's node boxes are (acceleration.rs:386-400,419-433).  The
// reference recurses to any depth; what it finds does not depend on the shape of the tree: a triangle that is hit lies
// in its leaf's box and so in every box above it, in any tree, and between equal distances the lower place in the
// leaf order wins, which is kept.  (A lopsided tree culls badly whatever walks it; a balanced one over the same
// leaves is no worse.)  Out: nodes in the layout of rt_mesh::nodes (first child = next record).
static void rebalance_tree(const rt_mesh& m, std::vector<rt_bvh_node>& out) {
    std::vector<uint32_t> leaves;
    {
        std::vector<uint32_t> st;
        st.push_back(0u);
        while (!st.empty()) {
            const uint32_t i = st.back(); st.pop_back();
            const rt_bvh_node& nd = m.nodes[i];
            if (nd.count != 0) { leaves.push_back(i); continue; }
            st.push_back(nd.right_or_first);
            st.push_back(i + 1);
        }
    }
    out.clear();
    out.reserve(leaves.size() * 2);
    std::function<uint32_t(size_t, size_t)> build = [&](size_t lo, size_t hi) -> uint32_t {
        const uint32_t me = (uint32_t)out.size();
        if (hi - lo == 1) { out.push_back(m.nodes[leaves[lo]]); return me; }
        out.emplace_back();
        const size_t mid = lo + (hi - lo + 1) / 2;
        build(lo, mid);                                  // lands at me + 1
        const uint32_t right = build(mid, hi);
        rt_bvh_node nd;
        for (int k = 0; k < 3; k++) {
            nd.min[k] = std::fmin(out[me + 1].min[k], out[right].min[k]);
            nd.max[k] = std::fmax(out[me + 1].max[k], out[right].max[k]);
        }
        nd.right_or_first = right; nd.count = 0;
        out[me] = nd;
        return me;
    };
    build(0, leaves.size());
}

int upload_mesh(rt_ctx* c, const rt_mesh& m, rt_ctx::MeshCopy& mc, bool rebalanced = false);
int upload_mesh(rt_ctx* c, const rt_mesh& m, rt_ctx::MeshCopy& mc, bool rebalanced) {
    if (!m.nodes && m.n_nodes == 0) return build_mesh_on_device(c, m, mc);
    DMesh& out = mc.dm;
    std::memset(&out, 0, sizeof(out));
    const bool dbg = std::getenv("RTGPU_DEBUG_BVH") != nullptr;
    auto tm0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  upload_mesh %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tm0).count());
        tm0 = t;
    };
    if (m.n_nodes == 0 || !m.nodes) return fail(c, RT_ERR_INVALID, "mesh without BVH nodes (the reference asserts sbvh.is_some(), mesh.rs:350)");
    if (m.n_tris == 0 || !m.tri_pos || !m.tri_nrm || !m.verts) return fail(c, RT_ERR_INVALID, "mesh without triangles");
    if (m.n_prims == 0 || !m.prim_indices) return fail(c, RT_ERR_INVALID, "mesh without leaf primitive list");
    if (m.n_prims >= 0x7FFFFFFFu) return fail(c, RT_ERR_UNSUPPORTED, "mesh too large");
    if (m.n_normals != 0 && !m.normals) return fail(c, RT_ERR_INVALID, "mesh: normals is NULL but n_normals is not 0");
    if (m.n_texcoords != 0 && !m.texcoords) return fail(c, RT_ERR_INVALID, "mesh: texcoords is NULL but n_texcoords is not 0");
    {
        const int rc_ = check_mesh_indices(c, m);
        if (rc_ != RT_OK) return rc_;
    }
    // walk the tree iteratively: validate links, number inner nodes, measure depth
    const uint32_t NN = m.n_nodes;
    std::vector<uint32_t> inner_id(NN, RT_NONE);
    std::vector<uint8_t> seen(NN, 0);
    uint32_t n_inner = 0, max_depth = 0;
    {
        std::vector<std::pair<uint32_t, uint32_t>> st;
        st.push_back({0u, 1u});
        while (!st.empty()) {
            auto [i, d] = st.back(); st.pop_back();
            if (i >= NN || seen[i]) return fail(c, RT_ERR_INVALID, "mesh BVH: bad child link");
            seen[i] = 1;
            if (d > max_depth) max_depth = d;
            const rt_bvh_node& nd = m.nodes[i];
            if (nd.count == 0) {
                inner_id[i] = 0;   // numbered below in index order (= DFS pre-order)
                st.push_back({nd.right_or_first, d + 1});
                st.push_back({i + 1, d + 1});
            } else {
                if ((uint64_t)nd.right_or_first + nd.count > m.n_prims) return fail(c, RT_ERR_INVALID, "mesh BVH: leaf range out of bounds");
            }
        }
    }
    for (uint32_t i = 0; i < m.n_prims; i++) if (m.prim_indices[i] >= m.n_tris) return fail(c, RT_ERR_INVALID, "mesh BVH: primitive index out of range");
    // too deep for the stacks (here, or for the packet walk's below): the same leaves under a balanced tree
    auto with_balanced_tree = [&]() {
        std::vector<rt_bvh_node> nn;
        rebalance_tree(m, nn);
        rt_mesh mm = m;
        mm.nodes = nn.data(); mm.n_nodes = (uint32_t)nn.size();
        if (dbg) std::fprintf(stderr, "  upload_mesh: tree of depth %u put under a balanced one (%zu nodes)\n", max_depth, nn.size());
        return upload_mesh(c, mm, mc, true);
    };
    if (max_depth > RT_MAX_STACK) {
        if (rebalanced) return fail(c, RT_ERR_UNSUPPORTED, "mesh BVH deeper than the traversal stack (" + std::to_string(max_depth) + " > " + std::to_string(RT_MAX_STACK) + ")");
        return with_balanced_tree();
    }
    for (uint32_t i = 0; i < NN; i++) if (seen[i] && m.nodes[i].count == 0) inner_id[i] = n_inner++;

    lap("validate");
    // f32 box of node i, rounded outward from the slab bounds the reference's test uses:
    // centroid -+ size with centroid = (min+max)*0.5 and size = max - centroid (acceleration.rs:76,135)
    auto box_of = [&](uint32_t i, float* bmin, float* bmax) {
        const rt_bvh_node& nd = m.nodes[i];
        for (int k = 0; k < 3; k++) {
            const double cen = (nd.min[k] + nd.max[k]) * 0.5;
            const double half = nd.max[k] - cen;
            const double lo = std::fmin(nd.min[k], cen - half), hi = std::fmax(nd.max[k], cen + half);
            float flo = (float)lo, fhi = (float)hi;
            if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
            if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
            bmin[k] = flo; bmax[k] = fhi;
        }
    };
    // Collapse to four-wide nodes: a node's children are its two reference children, the larger (by surface
    // area) inner ones replaced by their own two children until four slots are used.  Children keep the
    // reference's left-to-right order.  Device nodes are numbered in depth-first order of this wider tree.
    auto area_of = [&](uint32_t i) {
        const rt_bvh_node& nd = m.nodes[i];
        const double x = nd.max[0] - nd.min[0], y = nd.max[1] - nd.min[1], z = nd.max[2] - nd.min[2];
        return 2. * x * y + 2. * x * z + 2. * y * z;
    };
    std::vector<DNode> nodes;
    nodes.reserve(n_inner / 2 + 1);
    std::vector<uint8_t> last(m.n_prims, 0);
    uint32_t pstack_need = 0;
    for (uint32_t i = 0; i < NN; i++)
        if (seen[i] && m.nodes[i].count != 0) last[m.nodes[i].right_or_first + m.nodes[i].count - 1] = 1;
    if (m.nodes[0].count == 0) {
        // (reference node, device node index, entries on the packet walk's stack when the node is entered)
        struct Work { uint32_t ri, di, held; };
        std::vector<Work> work;
        nodes.emplace_back();
        std::memset(&nodes.back(), 0, sizeof(DNode));
        nodes.back().pad[0] = RT_NONE;
        work.push_back({0u, 0u, 0u});
        while (!work.empty()) {
            auto [ri, di, held] = work.back(); work.pop_back();
            uint32_t kids[4]; int nk = 2;
            kids[0] = ri + 1; kids[1] = m.nodes[ri].right_or_first;
            while (nk < 4) {
                int best = -1; double ba = -1.;
                for (int k = 0; k < nk; k++)
                    if (m.nodes[kids[k]].count == 0) { const double a = area_of(kids[k]); if (best < 0 || a > ba) { best = k; ba = a; } }
                if (best < 0) break;
                const uint32_t e = kids[best];
                for (int k = nk; k > best + 1; k--) kids[k] = kids[k - 1];
                kids[best] = e + 1; kids[best + 1] = m.nodes[e].right_or_first;
                nk++;
            }
            // topology now (sequential: the numbering depends on it); the boxes are filled in afterwards, in parallel.
            // Until then pad[k] remembers which reference node slot k shows.
            // (minx[k] holds, until the boxes are filled in, the number of the reference node slot k shows)
            DNode dn = nodes[di];              // pad[0], pad[1] (parent record, slot in it) were set when the parent was emitted
            uint32_t child_dev[4] = {0, 0, 0, 0};
            uint32_t n_inner_kids = 0;
            for (int k = 0; k < 4; k++) {
                uint32_t shown = RT_NONE;
                if (k >= nk) dn.ref[k] = RT_NO_CHILD;
                else {
                    shown = kids[k];
                    const rt_bvh_node& kn = m.nodes[kids[k]];
                    if (kn.count != 0) dn.ref[k] = RT_LEAF_BIT | kn.right_or_first;
                    else {
                        child_dev[k] = (uint32_t)nodes.size(); nodes.emplace_back(); dn.ref[k] = child_dev[k];
                        std::memset(&nodes.back(), 0, sizeof(DNode));
                        nodes.back().pad[0] = di; nodes.back().pad[1] = (uint32_t)k;
                        n_inner_kids++;
                    }
                }
                std::memcpy(&dn.minx[k], &shown, 4);
            }
            dn.pad[2] = n_inner_kids; dn.pad[3] = (uint32_t)nk;   // (the refit's arrival count; children in use)
            nodes[di] = dn;
            // the packet walks stack every child that was entered and take the top one
            pstack_need = std::max(pstack_need, held + (uint32_t)nk);
            for (int k = nk - 1; k >= 0; k--)
                if (m.nodes[kids[k]].count == 0) work.push_back({kids[k], child_dev[k], held + (uint32_t)nk - 1u});
        }
    }
    n_inner = (uint32_t)nodes.size();
    parallel_for(nodes.size(), [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            DNode& dn = nodes[i];
            for (int k = 0; k < 4; k++) {
                uint32_t shown;
                std::memcpy(&shown, &dn.minx[k], 4);
                if (shown == RT_NONE) {
                    // empty slot: a box no ray enters
                    dn.minx[k] = dn.miny[k] = dn.minz[k] = INFINITY;
                    dn.maxx[k] = dn.maxy[k] = dn.maxz[k] = -INFINITY;
                } else {
                    float bmin[3], bmax[3];
                    box_of(shown, bmin, bmax);
                    dn.minx[k] = bmin[0]; dn.miny[k] = bmin[1]; dn.minz[k] = bmin[2];
                    dn.maxx[k] = bmax[0]; dn.maxy[k] = bmax[1]; dn.maxz[k] = bmax[2];
                }
            }
        }
    });
    lap("collapse");
    if (pstack_need > RT_PSTACK) {
        if (rebalanced) return fail(c, RT_ERR_UNSUPPORTED, "mesh BVH needs " + std::to_string(pstack_need) + " packet stack entries (limit " + std::to_string(RT_PSTACK) + ")");
        return with_balanced_tree();
    }
    if (m.n_prims > RT_PACKET_MAX_PRIMS)
        return fail(c, RT_ERR_UNSUPPORTED, "mesh BVH with more than " + std::to_string(RT_PACKET_MAX_PRIMS) + " leaf entries");
    if (std::getenv("RTGPU_DEBUG_BVH")) std::fprintf(stderr, "binary depth %u, wide nodes %u, packet stack %u\n", max_depth, n_inner, pstack_need);
    std::unique_ptr<DTri[]> tris(new DTri[m.n_prims]);   // not zeroed: every field is written below, by the thread that first touches the page
    parallel_for(m.n_prims, [&](size_t a, size_t b) {
        for (size_t r = a; r < b; r++) {
            const uint32_t t = m.prim_indices[r];
            const double* v0 = &m.verts[3 * (size_t)m.tri_pos[3 * t]];
            const double* v1 = &m.verts[3 * (size_t)m.tri_pos[3 * t + 1]];
            const double* v2 = &m.verts[3 * (size_t)m.tri_pos[3 * t + 2]];
            DTri& dt = tris[r];
            for (int k = 0; k < 3; k++) { dt.v0[k] = v0[k]; dt.e1[k] = v1[k] - v0[k]; dt.e2[k] = v2[k] - v0[k]; }   // mesh.rs:26-27
            dt.orig = t; dt.last = last[r];
        }
    });
    if (!last[m.n_prims - 1]) return fail(c, RT_ERR_INVALID, "mesh BVH: leaf list does not end on a leaf boundary");
    lap("triangles");
    set_root_box(out, m.nodes[0].min, m.nodes[0].max);
    out.root_ref = m.nodes[0].count == 0 ? 0u : (RT_LEAF_BIT | m.nodes[0].right_or_first);
    out.n_inner = n_inner;
    int rc;
    if ((rc = upload(c, nodes.data(), nodes.size(), &out.nodes)) != RT_OK) return rc;
    if ((rc = upload(c, tris.get(), (size_t)m.n_prims, &out.tris)) != RT_OK) return rc;
    mc.n_records = m.n_prims;
    if ((rc = upload_mesh_attributes(c, m, mc)) != RT_OK) return rc;
    lap("copies");
    return RT_OK;
}

size_t level_bytes(size_t cap, bool keys) {
    // 6+2+3+3 doubles, 2 floats, 6 u32 per slot, and in scenes with area lights the ray's key; each array padded to 256 B
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return (keys ? 15 : 14) * pad(cap * 8) + 2 * pad(cap * 4) + 6 * pad(cap * 4);
}

int ensure_level(rt_ctx* c, size_t idx, size_t cap) {
    if (c->levels.size() <= idx) c->levels.resize(idx + 1);
    LevelAlloc& la = c->levels[idx];
    const bool keys = c->has_area;     // (only the kernels of scenes with area lights read or write a level's keys)
    if (la.cap >= cap && la.base && (la.keys || !keys)) return RT_OK;
    // A level that has to grow grows by an eighth more than asked: along a camera path the rays of a level drift up
    // and down by a per cent or so from frame to frame, and a buffer sized exactly would be freed and allocated again
    // (gigabytes, with the device idle meanwhile) for every frame that needs a few rays more than the last.
    if (la.base && la.cap > 0 && cap > la.cap) cap += cap / 8;
    if (la.base) { (void)hipFree(la.base); c->work_bytes -= level_bytes(la.cap, la.keys); la.base = nullptr; la.cap = 0; }
    size_t bytes = level_bytes(cap, keys);
    if (hipMalloc(&la.base, bytes) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes for a wavefront level failed");
    c->work_bytes += bytes;
    la.cap = cap; la.keys = keys;
    char* p = static_cast<char*>(la.base);
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto takeD = [&]() { double* r = reinterpret_cast<double*>(p); p += pad(cap * 8); return r; };
    auto takeF = [&]() { float* r = reinterpret_cast<float*>(p); p += pad(cap * 4); return r; };
    auto takeU = [&]() { uint32_t* r = reinterpret_cast<uint32_t*>(p); p += pad(cap * 4); return r; };
    DLevel& L = la.lv;
    for (int k = 0; k < 3; k++) L.ro[k] = takeD();
    for (int k = 0; k < 3; k++) L.rd[k] = takeD();
    L.hit_t = takeD(); L.hit_t2 = takeD();
    for (int k = 0; k < 3; k++) L.nrm[k] = takeD();
    // the lit colour takes the normal's place: k_light, the normal's last reader, is the colour's writer (same thread, same slot)
    for (int k = 0; k < 3; k++) L.lit[k] = L.nrm[k];
    L.w_t = takeD(); L.w_r = takeD(); L.emis = takeD();
    L.key = keys ? reinterpret_cast<unsigned long long*>(takeD()) : nullptr;
    L.hit_u = takeF(); L.hit_v = takeF();
    L.hit_obj = takeU(); L.hit_prim = takeU(); L.child0 = takeU(); L.child1 = takeU(); L.base_flags = takeU(); L.result = takeU();
    return RT_OK;
}

int ensure_buf(rt_ctx* c, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return RT_OK;
    if (*p && *cap > 0 && bytes > *cap) bytes += bytes / 8;     // (a buffer that grows: see ensure_level)
    if (*p) { (void)hipFree(*p); c->work_bytes -= *cap; *p = nullptr; *cap = 0; }
    if (hipMalloc(p, bytes) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc failed for an output buffer");
    *cap = bytes; c->work_bytes += bytes;
    return RT_OK;
}

// ---- half-precision floats on the host (the 64-byte object-tree records): value of a bit pattern, and the nearest
// patterns not above / not below a float (|x| below 65504)
float half_value(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    float f;
    if (e == 31u) { const uint32_t u = sign | 0x7F800000u | (m << 13); std::memcpy(&f, &u, 4); return f; }
    if (e == 0u) { f = std::ldexp((float)m, -24); return sign ? -f : f; }
    const uint32_t u = sign | ((e + 112u) << 23) | (m << 13);
    std::memcpy(&f, &u, 4);
    return f;
}
uint16_t half_step(uint16_t h, bool up) {   // the next pattern above (up) or below the value of h
    if (up) { if (h == 0x8000u) h = 0; return (h & 0x8000u) ? (uint16_t)(h - 1) : (uint16_t)(h + 1); }
    if (h == 0) h = 0x8000u;
    return (h & 0x8000u) ? (uint16_t)(h + 1) : (uint16_t)(h - 1);
}
uint16_t half_near(float x) {               // some pattern within one step of x (truncated mantissa)
    uint32_t u; std::memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 255u) - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7BFFu);
    if (e <= 0) return (uint16_t)sign;       // (tiny: +-0; the directed steps below take it from there)
    return (uint16_t)(sign | ((uint32_t)e << 10) | ((u >> 13) & 1023u));
}
uint16_t half_down(float x) {               // the largest half <= x
    uint16_t h = half_near(x);
    while (half_value(h) > x) h = half_step(h, false);
    while (half_value(half_step(h, true)) <= x) h = half_step(h, true);
    return h;
}
uint16_t half_up(float x) {                 // the smallest half >= x
    uint16_t h = half_near(x);
    while (half_value(h) < x) h = half_step(h, true);
    while (half_value(half_step(h, false)) >= x) h = half_step(h, false);
    return h;
}

// ---- object tree (DTlasNode): median split of the objects' boxes on the widest axis of their centres, emitted in
// depth-first order with skip links
struct TlasItem { double lo[3], hi[3]; uint32_t object; };
float f32_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; }
float f32_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; }
void build_tlas(std::vector<TlasItem>& items, size_t a, size_t b, std::vector<DTlasNode>& out) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = a; i < b; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], items[i].lo[k]); hi[k] = std::max(hi[k], items[i].hi[k]);
            const double ctr = 0.5 * (items[i].lo[k] + items[i].hi[k]);
            clo[k] = std::min(clo[k], ctr); chi[k] = std::max(chi[k], ctr);
        }
    const size_t me = out.size();
    out.emplace_back();
    for (int k = 0; k < 3; k++) {
        // inflated: the exact object tests are f64 on the ray as given, the boxes only have to contain them
        const double pad = 1e-5 * (std::fabs(lo[k]) + std::fabs(hi[k]) + (hi[k] - lo[k])) + 1e-30;
        out[me].bmin[k] = f32_down(lo[k] - pad); out[me].bmax[k] = f32_up(hi[k] + pad);
    }
    if (b - a == 1) { out[me].object = items[a].object; out[me].skip = (uint32_t)out.size(); return; }
    int axis = 0;
    for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    const size_t mid = a + (b - a) / 2;
    std::nth_element(items.begin() + (long)a, items.begin() + (long)mid, items.begin() + (long)b, [axis](const TlasItem& x, const TlasItem& y) {
        return x.lo[axis] + x.hi[axis] < y.lo[axis] + y.hi[axis];
    });
    build_tlas(items, a, mid, out);
    out[me].object = RT_TLAS_INNER | (uint32_t)out.size();
    build_tlas(items, mid, b, out);
    out[me].skip = (uint32_t)out.size();
}

// tiles per supertile side as a shift; one value per process (RTGPU_SUPER_SHIFT=0..4 overrides the default 2)
uint32_t super_shift() {
    static const uint32_t v = [] {
        const char* e = std::getenv("RTGPU_SUPER_SHIFT");
        const long x = e ? std::strtol(e, nullptr, 10) : 2;
        return (uint32_t)(x < 0 ? 0 : x > 4 ? 4 : x);
    }();
    return v;
}

// The rotation per row of supertiles (see ShardMap): with it the owner of column x in row y is (x + y * g) mod n for a
// step g near 0.38 n that is coprime to n.
uint32_t shard_row_rotation(uint32_t supers_x, uint32_t n_shards) {
    if (n_shards <= 1 || supers_x == 0) return 0;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
    uint32_t g = std::max<uint32_t>(1, (uint32_t)std::lround(0.382 * n_shards));
    while (gcd(g, n_shards) != 1) g++;
    // unrotated, the step is supers_x mod n; a rotation by R columns makes it (supers_x - R) mod n
    return (supers_x % n_shards + n_shards - g % n_shards) % n_shards;
}

ShardMap make_map(int32_t vw, int32_t vh, uint32_t shard, uint32_t n_shards, uint32_t row_major) {
    ShardMap m;
    m.shard = shard; m.n_shards = n_shards;
    m.super_shift = super_shift();
    const int32_t side = RT_TILE_W << m.super_shift;
    m.supers_x = (uint32_t)((vw + side - 1) / side);
    m.n_supers = m.supers_x * (uint32_t)((vh + side - 1) / side);
    m.row_rot = shard_row_rotation(m.supers_x, n_shards);
    m.row_major_out = row_major;
    m.whole_frame_slots = 0;
    const int64_t px = rt_shard_pixels(vw, vh, shard, n_shards);
    m.per_frame = m.frame_stride = px > 0 ? (uint32_t)px : 1u;
    return m;
}

uint64_t valid_pixels(const ShardMap& m, int32_t vw, int32_t vh) {
    uint64_t n = 0;
    const int32_t side = RT_TILE_W << m.super_shift;
    for (uint64_t sid = m.shard; sid < m.n_supers; sid += m.n_shards) {
        const int32_t sy = (int32_t)(sid / m.supers_x);
        const int32_t sx = (int32_t)((sid % m.supers_x + (uint64_t)sy * m.row_rot) % m.supers_x);
        n += (uint64_t)std::min<int32_t>(side, vw - sx * side) * (uint64_t)std::min<int32_t>(side, vh - sy * side);
    }
    return n;
}

int ensure_level_work(rt_ctx* c, size_t depth, size_t n) {
    if (c->lw.size() <= depth) c->lw.resize(depth + 1);
    LevelWork& W = c->lw[depth];
    for (auto& e : W.ev) if (!e) HIP_TRY(c, hipEventCreate(&e));
    int rc;
    if ((rc = ensure_buf(c, &W.hitlist, &W.hitlist_cap, n * 4)) != RT_OK) return rc;
    if (c->n_lights && (rc = ensure_buf(c, &W.occ, &W.occ_cap, n * c->n_lights)) != RT_OK) return rc;
    // long-ray lists: one entry per ray (per ray and light), only for launches small enough to use them
    if (n < c->long_below) {
        if (c->long_params.budget_lane && (rc = ensure_buf(c, &W.long_trace, &W.long_trace_cap, n * 4)) != RT_OK) return rc;
        if (c->long_params.budget_wave && c->n_lights && (rc = ensure_buf(c, &W.long_shadow, &W.long_shadow_cap, n * c->n_lights * 8)) != RT_OK) return rc;
    }
    return ensure_buf(c, &W.hs, &W.hs_cap, n * 48);
}

int ensure_level_counts(rt_ctx* c, size_t n) {
    if (c->n_lcounts >= n) return RT_OK;
    if (c->d_lcounts) (void)hipFree(c->d_lcounts);
    if (c->h_lcounts) (void)hipHostFree(c->h_lcounts);
    c->d_lcounts = nullptr; c->h_lcounts = nullptr; c->n_lcounts = 0;
    if (hipMalloc((void**)&c->d_lcounts, n * sizeof(DLevelCounts)) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipMalloc failed for the level counters");
    if (hipHostMalloc((void**)&c->h_lcounts, n * sizeof(DLevelCounts), hipHostMallocDefault) != hipSuccess) return fail(c, RT_ERR_NOMEM, "hipHostMalloc failed for the level counters");
    c->n_lcounts = n;
    return RT_OK;
}

// The frame: levels 0..max_depth forward, then folded back.
//
// Main stream:  k_primary, then per level d  k_trace(d) -> k_spawn(d).
// Side streams: per level d  k_shadow(d) -> k_light(d), both after k_spawn(d).
// The shadow casts of a level need only that level's hits, and the next level needs only the spawned rays, so the
// two chains run side by side: the long tail of a shadow launch (a few rays that skim the whole mesh) overlaps with
// the trace of the deeper levels instead of idling the chip.  k_resolve waits for every k_light.
//
// How many rays a level holds is only known on the device, once the level above has spawned them.  A frame can be
// rendered two ways:
//   measured    (learned == nullptr) before a level spawns, a small kernel counts the hits that can spawn (an upper
//               bound of the next level's size); the host reads that back (one stream synchronisation per level)
//               and gives the next level exactly that room;
//   speculative (learned = the level sizes of an earlier frame of this context and shard) every launch of the frame
//               is queued at once: level d is given room for a quarter more rays than it held last time (half more below
//               level 1), its launches
//               are sized for that room and read the real count from device memory.  If a level outgrows its room
//               the frame is rendered again the measured way (`overflow`); a sequence's frames resemble each other,
//               so that is rare.  No host round trip inside the frame: the launches of a small shard (a 1/8 frame is
//               1.5 ms of GPU work) are not held up by five waits for the host.
//
// n_frames > 1: the same shard of n_frames frames that differ in their camera only (d_cams, device memory), as one
// set of launches: level 0 holds the frames' primary rays one frame after the other, and nothing below level 0
// knows of frames.  Eight 1/8-frame shards make launches as large as a whole frame's, and a GPU that owns one shard
// of a sequence works as efficiently as a GPU that owns whole frames.
// pass_of_whole: the shard is one of n_shards passes over a whole frame (render_frame): compact output goes to the whole
// frame's slots, and frame_stride is the whole frame's.
int render_frame_impl(rt_ctx* c, uint32_t shard, uint32_t n_shards, uint32_t row_major, uint32_t n_frames, const DCamera* d_cams, uint32_t frame_stride,
                      uint8_t* d_rgb, uint32_t* d_ids, rt_stats* st,
                      const std::vector<uint64_t>* learned, bool& overflow, std::vector<uint64_t>& level_sizes, bool pass_of_whole) {
    overflow = false;
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t n0s = rt_shard_pixels(c->vw, c->vh, shard, n_shards);
    if (n0s < 0) return fail(c, RT_ERR_INVALID, "bad shard arguments");
    if ((uint64_t)n0s * n_frames >= 0x7FFFFFFFull) return fail(c, RT_ERR_UNSUPPORTED, "frame (or batch of frames) too large for one set of launches");
    const uint32_t per_frame = (uint32_t)n0s;
    const uint32_t n0 = per_frame * n_frames;
    ShardMap map = make_map(c->vw, c->vh, shard, n_shards, row_major);
    if (n_frames > 1) map.frame_stride = frame_stride;
    if (pass_of_whole) {
        map.whole_frame_slots = 1;
        if (n_frames == 1) map.frame_stride = (uint32_t)rt_shard_pixels(c->vw, c->vh, 0, 1);
    }
    rt_stats stats; std::memset(&stats, 0, sizeof(stats));
    stats.rays_primary = valid_pixels(map, c->vw, c->vh) * n_frames;
    stats.rays_per_level[0] = stats.rays_primary;
    hipStream_t s = c->stream;
    int rc;
    const size_t n_lc = (size_t)c->max_depth + 2;
    if (n_lc > RT_MAX_LEVELS) return fail(c, RT_ERR_UNSUPPORTED, "max_ray_depth above 31 is not supported");
    if ((rc = ensure_level_counts(c, n_lc)) != RT_OK) return rc;
    // (only the counters of the levels this scene can have are cleared, copied back and summed)
    std::memset(c->hcounters, 0, n_lc * sizeof(DCounters));
    HIP_TRY(c, hipMemsetAsync(c->dcounters, 0, n_lc * sizeof(DCounters), s));
    HIP_TRY(c, hipMemsetAsync(c->d_lcounts, 0, n_lc * sizeof(DLevelCounts), s));

    // room per level: exact for level 0; measured frames learn the others as they go
    std::vector<uint32_t> room;
    room.push_back(n0);
    if (learned) {
        for (size_t d = 1; d < learned->size() && d <= c->max_depth; d++) {
            const uint64_t worst = std::min<uint64_t>(2ull * room[d - 1], 0x7FFFFFFEull);
            // level 1 follows the picture closely from frame to frame; the deeper levels are small and jumpy (rays
            // through a glass sphere that a few degrees of camera turn in or out of total reflection: +39 % from one
            // set of ten frames to the next in scenes/refraction.sdl), and room for them costs little
            const uint64_t want = d == 1 ? (*learned)[d] + (*learned)[d] / 4 + 4096 : (*learned)[d] + (*learned)[d] / 2 + 16384;
            room.push_back((uint32_t)std::min(worst, want));
        }
    }
    std::vector<uint8_t> main_shadow(RT_MAX_LEVELS, 0);   // levels whose shadow casts ran as the budget-free packet kernel
    uint32_t n_tr = 0, n_sh = 0, n_sd = 0, n_main = 0;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    if (n0 > 0) {
        if ((rc = ensure_level(c, 0, n0)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_primary, dim3((n0 + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, s, c->dscene, d_cams, c->levels[0].lv, n0, map);
    }
    for (uint32_t depth = 0; depth < room.size(); depth++) {
        const uint32_t n = room[depth];   // rays of this level (measured) or the room given to it (speculative)
        if (n == 0) break;
        // the level's real size, for the kernels: level 0's is n, the others' stands in device memory
        const unsigned long long* n_dev = depth > 0 ? &c->d_lcounts[depth - 1].next_count : nullptr;
        if ((rc = ensure_level_work(c, depth, n)) != RT_OK) return rc;
        LevelWork& W = c->lw[depth];
        double* hs = static_cast<double*>(W.hs);
        uint32_t* hitlist = static_cast<uint32_t*>(W.hitlist);
        uint8_t* occ = static_cast<uint8_t*>(W.occ);
        DLevelCounts* lc = c->d_lcounts + depth;
        DCounters* dc = c->dcounters + depth;
        uint32_t* long_trace = static_cast<uint32_t*>(W.long_trace);
        uint32_t* long_shadow = static_cast<uint32_t*>(W.long_shadow);
        const unsigned long long* keys = c->levels[depth].lv.key;   // (ensure_level(depth) ran before this level was spawned into)
        // budgets only where the launch is small enough for its slowest wave to set its duration: in a large
        // launch the other waves hide a long walk, and cutting it up would only add work
        LongParams lp = c->long_params;
        if (n >= c->long_below) { lp.budget_lane = 0; lp.budget_wave = 0; }
        hipStream_t side = c->serial ? s : c->side[depth % c->n_side];
        const dim3 grid((n + RT_BLOCK - 1) / RT_BLOCK), block(RT_BLOCK);
        HIP_TRY(c, hipEventRecord(W.ev[LW_T0], s));
        // depth 0 of a scene with meshes: primary rays, 8x8 pixel blocks per wave -> coherent packets; deeper rays
        // (and scenes of analytic objects only, whose object tree is walked per lane anyway) lane by lane
        // (a scene of analytic objects only has no plain packets: its object tree is walked by the deferred packets at
        // every level, or lane by lane)
        const bool object_tree_only = c->hscene.n_meshes == 0 && c->hscene.n_tlas != 0 && c->defer_from <= RT_MAX_LEVELS && (uint32_t)depth < c->defer_objects_below;
        // Level 0 as well where triangles are about as small as pixels (one per 16 pixels or more): the lanes of a
        // primary packet then end in different leaves, and testing a leaf with the whole wave is what the queues avoid
        // (config 4: 2 % of the frame; on the coarse meshes of the other configurations plain packets are 5-10 % faster)
        const bool fine_mesh = !c->defer_from_set && c->total_tris * 16 >= (uint64_t)c->vw * (uint64_t)c->vh;
        const bool deferred = object_tree_only ||
                              (c->hscene.n_meshes != 0 && !c->any_need_far && (depth >= c->defer_from || fine_mesh) && (!lp.budget_lane || c->defer_small));
        if (deferred) lp.budget_lane = 0;
        {
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, grid, block, 0, s, c->dscene, c->levels[depth].lv, n, n_dev, hitlist, hs, n, dc, lc, long_trace, lp);
            };
            const bool cnt = (c->count_visits & 1) != 0;
            if (deferred) { if (cnt) launch(k_trace_deferred<true>); else if (c->rays_need_reciprocal) launch(k_trace_deferred<false>); else launch(k_trace_deferred<false, false>); }
            else if (c->hscene.n_meshes == 0) { if (cnt) launch(k_trace<true, false, false, false, false>); else launch(k_trace<false, false, false, false, false>); }
            else if (depth <= c->packet_depth) {   // (packets pay for mesh walks only)
                if (c->any_need_far) { if (cnt) launch(k_trace_packet<true, true>); else launch(k_trace_packet<false, true>); }
                else { if (cnt) launch(k_trace_packet<true, false>); else launch(k_trace_packet<false, false>); }
            }
            else if (lp.budget_lane) { if (cnt) launch(k_trace<true, false, true>); else launch(k_trace<false, false, true>); }
            else if (c->any_need_far) { if (cnt) launch(k_trace<true, false, false, true>); else launch(k_trace<false, false, false, true>); }
            else { if (cnt) launch(k_trace<true, false, false, false>); else launch(k_trace<false, false, false, false>); }
        }
        if (depth > 0 && lp.budget_lane) {
            if (c->count_visits & 1) hipLaunchKernelGGL(k_long_trace<true>, dim3(RT_LONG_BLOCKS), block, 0, s, c->dscene, c->levels[depth].lv, long_trace, hitlist, hs, n, dc, lc);
            else hipLaunchKernelGGL(k_long_trace<false>, dim3(RT_LONG_BLOCKS), block, 0, s, c->dscene, c->levels[depth].lv, long_trace, hitlist, hs, n, dc, lc);
        }
        HIP_TRY(c, hipEventRecord(W.ev[LW_T1], s));
        // Room for the next level.  Speculative: what it needed last time and a quarter.  Measured: a count of the
        // hits that can spawn (one small kernel and one read-back per level), which bounds it from above.
        size_t next_cap = 0;
        if (learned) next_cap = depth + 1 < room.size() ? room[depth + 1] : 0;
        else if (depth < c->max_depth) {
            hipLaunchKernelGGL(k_count_children, grid, block, 0, s, c->dscene, c->levels[depth].lv, n, n_dev, depth, lc);
            HIP_TRY(c, hipMemcpyAsync(c->h_lcounts + depth, lc, sizeof(DLevelCounts), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, hipGetLastError());
            next_cap = (size_t)std::min<unsigned long long>(c->h_lcounts[depth].child_bound, 0x7FFFFFFEull);
            if (next_cap > 0) room.push_back((uint32_t)next_cap);
        }
        if ((rc = ensure_level(c, depth + 1, std::max<size_t>(next_cap, 1))) != RT_OK) return rc;
        {
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, grid, block, 0, s, c->dscene, c->levels[depth].lv, c->levels[depth + 1].lv, n, n_dev, depth,
                                   (uint32_t)next_cap, dc, lc, depth == 0 ? d_ids : nullptr, map);
            };
            const bool cnt = (c->count_visits & 4) != 0;
            if (c->any_mesh_refracts) { if (cnt) launch(k_spawn<true, true>); else launch(k_spawn<false, true>); }
            else { if (cnt) launch(k_spawn<true, false>); else launch(k_spawn<false, false>); }
        }
        HIP_TRY(c, hipEventRecord(W.ev[LW_P], s));
        // the shadow casts start after k_spawn, not beside it: the spawn is short but on the frame's critical
        // path, and next to a chip full of shadow waves it would crawl
        HIP_TRY(c, hipStreamWaitEvent(side, W.ev[LW_P], 0));
        HIP_TRY(c, hipEventRecord(W.ev[LW_S0], side));
        bool main_variant = false;
        if (c->n_lights) {
            // sized for the worst case (every ray hit); threads past the hit count leave at once
            const dim3 sgrid((n + RT_BLOCK - 1) / RT_BLOCK, c->n_lights);
            if (c->hscene.n_meshes == 0 && !(c->hscene.n_tlas != 0 && c->defer_from <= RT_MAX_LEVELS)) {
                lp.budget_wave = 0;   // (no long-ray pass: nothing is walked as a packet)
                hipLaunchKernelGGL(k_shadow_lanes<false>, sgrid, block, 0, side, c->dscene, hitlist, hs, occ, n, dc, lc, keys);
            } else {
                auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, sgrid, block, 0, side, c->dscene, hitlist, hs, occ, n, dc, lc, long_shadow, lp); };
                auto launch_area = [&](auto kern) { hipLaunchKernelGGL(kern, sgrid, block, 0, side, c->dscene, hitlist, hs, occ, n, dc, lc, long_shadow, lp, keys); };
                const bool cnt = (c->count_visits & 2) != 0;
                const bool rcp = c->rays_need_reciprocal;
                if (lp.budget_wave) {
                    if (c->has_area) { if (cnt) launch_area(k_shadow_area<true, true>); else launch_area(k_shadow_area<false, true>); }
                    else if (cnt) launch(k_shadow<true, true>);
                    else if (rcp) launch(k_shadow<false, true>); else launch(k_shadow<false, true, false>);
                } else {
                    main_variant = true;
                    if (c->has_area) { if (cnt) launch_area(k_shadow_area<true, false>); else launch_area(k_shadow_area<false, false>); }
                    else if (cnt) launch(k_shadow<true, false>);
                    else if (rcp) launch(k_shadow<false, false>); else launch(k_shadow<false, false, false>);
                }
            }
        }
        HIP_TRY(c, hipEventRecord(W.ev[LW_SK], side));
        if (c->n_lights) {
            if (lp.budget_wave) {
                if (c->count_visits & 2) hipLaunchKernelGGL(k_long_shadow<true>, dim3(RT_LONG_BLOCKS), block, 0, side, c->dscene, hitlist, hs, occ, n, dc, lc, long_shadow, keys);
                else hipLaunchKernelGGL(k_long_shadow<false>, dim3(RT_LONG_BLOCKS), block, 0, side, c->dscene, hitlist, hs, occ, n, dc, lc, long_shadow, keys);
            }
            n_sd++;
        }
        if (main_variant) { main_shadow[depth] = 1; n_main++; }
        HIP_TRY(c, hipEventRecord(W.ev[LW_S1], side));
        HIP_TRY(c, hipEventRecord(W.ev[LW_L0], side));
        if (c->has_area) hipLaunchKernelGGL(k_light_area, grid, block, 0, side, c->dscene, c->levels[depth].lv, n, n_dev, occ, n, dc);
        else hipLaunchKernelGGL(k_light, grid, block, 0, side, c->dscene, c->levels[depth].lv, n, n_dev, occ, n);
        HIP_TRY(c, hipEventRecord(W.ev[LW_L1], side));
        n_tr++; n_sh++;
    }
    size_t n_levels = 0;
    while (n_levels < room.size() && room[n_levels] > 0) n_levels++;
    const size_t used = std::min<size_t>(n_levels, c->lw.size());
    for (size_t d = 0; d < used; d++) HIP_TRY(c, hipStreamWaitEvent(s, c->lw[d].ev[LW_L1], 0));
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    for (int depth = (int)n_levels - 1; depth >= 0; depth--) {
        const uint32_t n = room[depth];
        const unsigned long long* n_dev = depth > 0 ? &c->d_lcounts[depth - 1].next_count : nullptr;
        hipLaunchKernelGGL(k_resolve, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, s, c->dscene, c->levels[depth].lv,
                           c->levels[depth + 1].lv, n, n_dev, depth == 0 ? 1 : 0, d_rgb, map);
    }
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    HIP_TRY(c, hipMemcpyAsync(c->hcounters, c->dcounters, n_lc * sizeof(DCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_lcounts, c->d_lcounts, n_lc * sizeof(DLevelCounts), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    // what the levels really held; a speculative frame whose level outgrew its room has dropped rays and is void
    level_sizes.assign(1, n0);
    for (size_t d = 0; d + 1 < n_lc; d++) {
        const unsigned long long spawned = c->h_lcounts[d].next_count;
        if (spawned == 0 || d >= n_levels) { if (spawned != 0) overflow = true; break; }
        const unsigned long long had_room = d + 1 < room.size() ? room[d + 1] : 0;
        if (spawned > had_room) overflow = true;
        level_sizes.push_back(spawned);
        if (d + 1 < 8) stats.rays_per_level[d + 1] = spawned;
    }
    if (overflow) return RT_OK;
    for (size_t d = 0; d < n_levels && d < n_lc; d++) { stats.long_trace += c->h_lcounts[d].long_trace; stats.long_shadow += c->h_lcounts[d].long_shadow; }
    // per-phase times are sums of the launches' own durations; the phases overlap, so they add up to more than ms_render
    float ms_trace = 0.f, ms_shadow = 0.f, ms_shade = 0.f, ms_resolve = 0.f, ms_all = 0.f, ms_main = 0.f;
    for (size_t d = 0; d < used; d++) {
        const LevelWork& W = c->lw[d];
        float a = 0.f, b = 0.f, p = 0.f, l = 0.f, k = 0.f;
        (void)hipEventElapsedTime(&a, W.ev[LW_T0], W.ev[LW_T1]);
        (void)hipEventElapsedTime(&b, W.ev[LW_S0], W.ev[LW_S1]);
        (void)hipEventElapsedTime(&p, W.ev[LW_T1], W.ev[LW_P]);
        (void)hipEventElapsedTime(&l, W.ev[LW_L0], W.ev[LW_L1]);
        if (main_shadow[d]) { (void)hipEventElapsedTime(&k, W.ev[LW_S0], W.ev[LW_SK]); ms_main += k; }
        ms_trace += a; ms_shadow += b; ms_shade += p + l;
    }
    (void)hipEventElapsedTime(&ms_resolve, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&ms_all, c->ev[0], c->ev[2]);
    for (size_t d = 0; d < n_lc; d++)
        for (int k = 0; k < RT_SHARDS; k++) {
            const DShard& sh = c->hcounters[d].shard[k];
            stats.rays_reflect += sh.rays_reflect; stats.rays_transmit += sh.rays_transmit; stats.rays_shadow += sh.rays_shadow;
            stats.nodes_visited += sh.nodes_visited; stats.tris_tested += sh.tris_tested;
            stats.wave_nodes += sh.wave_nodes; stats.wave_tris += sh.wave_tris;
            stats.bytes_fetched += 16ull * sh.units_fetched;
            if (main_shadow[d]) stats.bytes_fetched_main += 16ull * sh.units_fetched;
            stats.max_lane_steps = std::max<uint64_t>(stats.max_lane_steps, sh.max_lane_steps); stats.max_wave_steps = std::max<uint64_t>(stats.max_wave_steps, sh.max_wave_steps);
        }
    stats.ms_render = ms_all; stats.ms_trace = ms_trace; stats.ms_shadow = ms_shadow; stats.ms_shade = ms_shade; stats.ms_resolve = ms_resolve;
    stats.ms_shadow_main = ms_main; stats.launches_shadow_main = n_main;
    stats.launches_trace = n_tr; stats.launches_shadow = n_sd; stats.launches_shade = n_sh;
    stats.speculative = learned ? 1u : 0u;
    stats.ms_upload = c->ms_upload;
    stats.device_bytes = c->scene_bytes + c->work_bytes;
    if (st) *st = stats;
    return RT_OK;
}

// Every error exit of a frame passes through here: kernels of earlier levels may still be writing the level buffers,
// the occlusion bytes or the caller's output buffer on the main and side streams, and the caller may free that
// buffer as soon as it sees the error code.
int render_shard_once(rt_ctx* c, uint32_t shard, uint32_t n_shards, uint32_t row_major, uint8_t* d_rgb, uint32_t* d_ids, rt_stats* st,
                      uint32_t n_frames, const DCamera* d_cams, uint32_t frame_stride, bool pass_of_whole) {
    // the level sizes this context last saw for this shard of this frame size, per frame: the next frame (or set of
    // frames: a set of n is given n times the room) is queued without waiting for its own (see render_frame_impl);
    // RTGPU_SYNC_LEVELS=1 turns that off
    const rt_ctx::LearnedKey key{shard, n_shards, c->vw, c->vh, c->max_depth, 0u};
    auto it = c->sync_levels ? c->learned.end() : c->learned.find(key);
    bool overflow = false;
    std::vector<uint64_t> sizes, scaled;
    if (it != c->learned.end()) { scaled = it->second; for (auto& v : scaled) v *= n_frames; }
    int rc = render_frame_impl(c, shard, n_shards, row_major, n_frames, d_cams, frame_stride, d_rgb, d_ids, st, it != c->learned.end() ? &scaled : nullptr, overflow, sizes, pass_of_whole);
    if (rc == RT_OK && overflow) {
        c->respeculated++;
        rc = render_frame_impl(c, shard, n_shards, row_major, n_frames, d_cams, frame_stride, d_rgb, d_ids, st, nullptr, overflow, sizes, pass_of_whole);
        if (rc == RT_OK && overflow) rc = fail(c, RT_ERR_DEVICE, "wavefront level overflow");
    }
    if (rc == RT_OK) {
        for (auto& v : sizes) v = (v + n_frames - 1) / n_frames;
        c->learned[key] = sizes;
        if (st) st->respeculated = c->respeculated;
    }
    if (rc != RT_OK) {
        const std::string keep = c->err;
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (auto side_s : c->side) if (side_s) (void)hipStreamSynchronize(side_s);
        (void)hipGetLastError();
        c->err = keep;
    }
    return rc;
}

// Passes over a whole frame.  The wavefront levels of a frame take memory in proportion to its rays (8K,
// scenes/huge_grid.sdl: 377 M rays in five levels, 115 GB), and a chip is full long before that: a whole frame (or set
// of whole frames) whose levels would take more than RT_PASS_BUDGET is rendered as P shards one after the other through
// the same level buffers, every pass writing its pixels where the whole frame's launches would have written them.
// The levels' size is known from the last such frame of the scene (about RT_PASS_BYTES_PER_RAY with the per-level work
// lists and the room levels are given); before there is one, frames of more than two 4K frames' pixels go in passes of
// one 4K frame's.  Passes cost a light frame time (8K terrain, 2.4 rays per pixel: 6.9 ms whole, 7.5 in two passes,
// 8.1 in four) and a heavy one nothing (huge_grid: 119.6 / 119.7 / 120.4 ms).
// RTGPU_PASSES=P in the environment fixes P (1: never).
#define RT_PASS_PIXELS (3840ull * 2160ull)
#define RT_PASS_BUDGET (32ull << 30)
#define RT_PASS_BYTES_PER_RAY 290ull
uint32_t passes_for(const rt_ctx* c, uint32_t n_frames) {
    const uint64_t supers = (uint64_t)rt_shard_pixels(c->vw, c->vh, 0, 1) >> (8u + 2u * super_shift());
    uint64_t p = c->passes;
    if (p == 0) {
        const auto it = c->whole_frame_rays.find(std::make_tuple(c->vw, c->vh, c->max_depth));
        if (it != c->whole_frame_rays.end()) {
            const uint64_t bytes = it->second * n_frames * RT_PASS_BYTES_PER_RAY;
            p = (bytes + RT_PASS_BUDGET - 1) / RT_PASS_BUDGET;
        } else {
            const uint64_t px = (uint64_t)c->vw * (uint64_t)c->vh * n_frames;
            p = px > 2 * RT_PASS_PIXELS ? (px + RT_PASS_PIXELS - 1) / RT_PASS_PIXELS : 1;
        }
    }
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({p, supers, 64}));
}

int render_frame(rt_ctx* c, uint32_t shard, uint32_t n_shards, uint32_t row_major, uint8_t* d_rgb, uint32_t* d_ids, rt_stats* st,
                 uint32_t n_frames = 1, const DCamera* d_cams = nullptr, uint32_t frame_stride = 0) {
    const uint32_t passes = n_shards == 1 && c->have_scene ? passes_for(c, n_frames) : 1;
    auto note_rays = [&](const rt_stats& s) {
        uint64_t rays = 0;
        for (int d = 0; d < 8; d++) rays += s.rays_per_level[d];
        c->whole_frame_rays[std::make_tuple(c->vw, c->vh, c->max_depth)] = (rays + n_frames - 1) / n_frames;
    };
    if (passes == 1) {
        rt_stats one;
        const int rc = render_shard_once(c, shard, n_shards, row_major, d_rgb, d_ids, &one, n_frames, d_cams, frame_stride, false);
        if (rc == RT_OK && n_shards == 1) note_rays(one);
        if (rc == RT_OK && st) *st = one;
        return rc;
    }
    rt_stats sum; std::memset(&sum, 0, sizeof(sum));
    for (uint32_t p = 0; p < passes; p++) {
        rt_stats one;
        const int rc = render_shard_once(c, p, passes, row_major, d_rgb, d_ids, &one, n_frames, d_cams, frame_stride, true);
        if (rc != RT_OK) return rc;
        // the passes run one after the other: rays and times add up
        sum.rays_primary += one.rays_primary; sum.rays_reflect += one.rays_reflect; sum.rays_transmit += one.rays_transmit; sum.rays_shadow += one.rays_shadow;
        for (int d = 0; d < 8; d++) sum.rays_per_level[d] += one.rays_per_level[d];
        sum.ms_render += one.ms_render; sum.ms_trace += one.ms_trace; sum.ms_shadow += one.ms_shadow; sum.ms_shade += one.ms_shade;
        sum.ms_resolve += one.ms_resolve; sum.ms_shadow_main += one.ms_shadow_main;
        sum.nodes_visited += one.nodes_visited; sum.tris_tested += one.tris_tested; sum.wave_nodes += one.wave_nodes; sum.wave_tris += one.wave_tris;
        sum.launches_trace += one.launches_trace; sum.launches_shadow += one.launches_shadow; sum.launches_shade += one.launches_shade;
        sum.launches_shadow_main += one.launches_shadow_main;
        sum.long_trace += one.long_trace; sum.long_shadow += one.long_shadow; sum.bytes_fetched += one.bytes_fetched; sum.bytes_fetched_main += one.bytes_fetched_main;
        sum.max_lane_steps = std::max(sum.max_lane_steps, one.max_lane_steps); sum.max_wave_steps = std::max(sum.max_wave_steps, one.max_wave_steps);
        sum.speculative = p == 0 ? one.speculative : (sum.speculative && one.speculative ? 1u : 0u);
        sum.ms_upload = one.ms_upload; sum.device_bytes = one.device_bytes; sum.respeculated = one.respeculated;
    }
    note_rays(sum);
    {
        // the whole frame's level sizes are the sums of the passes': should the next frame go in one pass (a light frame,
        // once that is known), it is queued with them like any other
        std::vector<uint64_t> whole;
        for (uint32_t p = 0; p < passes; p++) {
            const auto it = c->learned.find(rt_ctx::LearnedKey{p, passes, c->vw, c->vh, c->max_depth, 0u});
            if (it == c->learned.end()) continue;
            if (whole.size() < it->second.size()) whole.resize(it->second.size(), 0);
            for (size_t d = 0; d < it->second.size(); d++) whole[d] += it->second[d];
        }
        if (!whole.empty()) c->learned[rt_ctx::LearnedKey{0u, 1u, c->vw, c->vh, c->max_depth, 0u}] = whole;
    }
    if (st) *st = sum;
    return RT_OK;
}

}  // namespace

extern "C" {

int64_t rt_shard_pixels(int32_t vw, int32_t vh, uint32_t shard, uint32_t n_shards) {
    if (vw <= 0 || vh <= 0 || n_shards == 0 || shard >= n_shards) return -1;
    const uint32_t sh = super_shift();
    const int32_t side = RT_TILE_W << sh;
    const uint64_t supers = (uint64_t)((vw + side - 1) / side) * (uint64_t)((vh + side - 1) / side);
    const uint64_t mine = shard < supers ? (supers - shard + n_shards - 1) / n_shards : 0;
    return (int64_t)((mine << (2 * sh)) * RT_TILE_W * RT_TILE_H);
}

uint32_t rt_supertile(void) { return 1u << super_shift(); }

int rt_ctx_create(int device_id, rt_ctx** out) {
    if (!out) { g_create_error = "rt_ctx_create: out is NULL"; return RT_ERR_INVALID; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"); return RT_ERR_DEVICE; }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return RT_ERR_INVALID; }
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) { g_create_error = "out of host memory"; return RT_ERR_NOMEM; }
    c->device = device_id;
    {
        // "1": every kernel instrumented; "trace" / "shadow" / "shade": only that one (to attribute the counts)
        const char* e = std::getenv("RTGPU_COUNT_VISITS");
        const std::string v = e ? e : "";
        c->count_visits = v == "1" ? 7 : v == "trace" ? 1 : v == "shadow" ? 2 : v == "shade" ? 4 : 0;
        if (const char* b = std::getenv("RTGPU_BUDGET_LANE")) c->long_params.budget_lane = (uint32_t)std::strtoul(b, nullptr, 10);
        if (const char* b = std::getenv("RTGPU_BUDGET_WAVE")) c->long_params.budget_wave = (uint32_t)std::strtoul(b, nullptr, 10);
        c->serial = std::getenv("RTGPU_SERIAL") != nullptr;
        if (const char* b = std::getenv("RTGPU_SYNC_LEVELS")) c->sync_levels = std::strtol(b, nullptr, 10) != 0;
        if (const char* b = std::getenv("RTGPU_COPY_THREADS")) c->copy_threads = (uint32_t)std::min<unsigned long>(std::strtoul(b, nullptr, 10), 16ul);
        if (const char* b = std::getenv("RTGPU_PACKET_DEPTH")) c->packet_depth = (uint32_t)std::strtoul(b, nullptr, 10);
        if (const char* b = std::getenv("RTGPU_DEFER_FROM")) { c->defer_from = (uint32_t)std::strtoul(b, nullptr, 10); c->defer_from_set = true; }
        if (const char* b = std::getenv("RTGPU_DEFER_SMALL")) c->defer_small = b[0] != '0';
        if (const char* b = std::getenv("RTGPU_DEFER_OBJECTS_BELOW")) c->defer_objects_below = (uint32_t)std::strtoul(b, nullptr, 10);
        if (const char* b = std::getenv("RTGPU_BUDGET_BELOW")) c->long_below = (uint32_t)std::strtoul(b, nullptr, 10);
        if (const char* b = std::getenv("RTGPU_PASSES")) c->passes = (uint32_t)std::strtoul(b, nullptr, 10);
    }
    for (auto& ev : c->ev) ev = nullptr;
    auto bail = [&](const char* what, hipError_t err) { g_create_error = std::string(what) + ": " + hipGetErrorString(err); rt_ctx_destroy(c); return RT_ERR_DEVICE; };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = scratch_address(device_id, &c->scratch)) != hipSuccess) return bail("hipGetSymbolAddress", e);
    {
        // A test knob, and the only state of the library that is not a context's own: how many entries of the packets'
        // node queue the walks use is a variable of the device image, shared by every context on that device.  It is
        // written only when RTGPU_NODE_QUEUE is set (tests make the queue overflow with it) or to undo such a write;
        // in a process that never sets the variable, creating a context changes nothing outside that context.
        static std::mutex mu;
        static bool lowered[RT_MAX_DEVICES] = {};
        std::lock_guard<std::mutex> lock(mu);
        const char* b = std::getenv("RTGPU_NODE_QUEUE");
        if (b || (device_id < RT_MAX_DEVICES && lowered[device_id])) {
            uint32_t v = b ? (uint32_t)std::strtoul(b, nullptr, 10) : (uint32_t)RT_NCAP;
            v = v < 4u ? 4u : (v > (uint32_t)RT_NCAP ? (uint32_t)RT_NCAP : v);
            if ((e = hipMemcpyToSymbol(HIP_SYMBOL(rt_node_cap), &v, sizeof(v))) != hipSuccess) return bail("hipMemcpyToSymbol", e);
            if (device_id < RT_MAX_DEVICES) lowered[device_id] = v != (uint32_t)RT_NCAP;
        }
    }
    // (all streams at the default priority: mixed priorities hung the frame on this ROCm)
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    // Queued all at once (no level sizes read back), a frame's five shadow chains wait in the hardware queues for events
    // far ahead and hold up whatever shares those queues: two side streams then serve better than four (4 % on
    // the whole frame with four frames in flight).
    c->n_side = c->sync_levels ? RT_SIDE_STREAMS : 2;
    if (const char* b = std::getenv("RTGPU_SIDE_STREAMS")) c->n_side = std::min<uint32_t>(RT_SIDE_STREAMS, std::max<uint32_t>(1, (uint32_t)std::strtoul(b, nullptr, 10)));
    for (uint32_t k = 0; k < c->n_side; k++) if ((e = hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipMalloc((void**)&c->dscene, sizeof(DScene))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->dcounters, RT_MAX_LEVELS * sizeof(DCounters))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc((void**)&c->hcounters, RT_MAX_LEVELS * sizeof(DCounters), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    for (auto& ev : c->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    *out = c;
    return RT_OK;
}

static void multi_destroy(rt_ctx* head);
void rt_ctx_destroy(rt_ctx* c) {
    if (!c) return;
    if (c->multi || c->solo) multi_destroy(c);   // the other members, the communicators and the gather buffers; then this context like any other
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto ss : c->side) if (ss) (void)hipStreamSynchronize(ss);
    free_scene(c);
    free_mesh_copies(c->mesh_copies);
    for (auto& la : c->levels) if (la.base) (void)hipFree(la.base);
    for (auto& W : c->lw) {
        if (W.hitlist) (void)hipFree(W.hitlist);
        if (W.occ) (void)hipFree(W.occ);
        if (W.hs) (void)hipFree(W.hs);
        if (W.long_trace) (void)hipFree(W.long_trace);
        if (W.long_shadow) (void)hipFree(W.long_shadow);
        for (auto ev : W.ev) if (ev) (void)hipEventDestroy(ev);
    }
    if (c->d_lcounts) (void)hipFree(c->d_lcounts);
    if (c->h_lcounts) (void)hipHostFree(c->h_lcounts);
    for (auto ss : c->side) if (ss) (void)hipStreamDestroy(ss);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->d_ids) (void)hipFree(c->d_ids);
    if (c->d_ids2) (void)hipFree(c->d_ids2);
    if (c->d_cams) (void)hipFree(c->d_cams);
    if (c->dscene) (void)hipFree(c->dscene);
    if (c->dcounters) (void)hipFree(c->dcounters);
    if (c->hcounters) (void)hipHostFree(c->hcounters);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static int scene_upload(rt_ctx* c, const rt_scene_desc* d);
static int upload_one(rt_ctx* c, const rt_scene_desc* d);
static int multi_upload(rt_ctx* head, const rt_scene_desc* d);
int rt_scene_upload(rt_ctx* c, const rt_scene_desc* d) {
    if (!c) return RT_ERR_INVALID;
    return c->multi ? multi_upload(c, d) : upload_one(c, d);
}
static int upload_one(rt_ctx* c, const rt_scene_desc* d) {
    const int rc = scene_upload(c, d);
    if (rc != RT_OK) {   // a failed upload leaves nothing behind, kept meshes included
        (void)hipSetDevice(c->device);
        free_scene(c);
        free_mesh_copies(c->mesh_copies);
    }
    return rc;
}
static int scene_upload(rt_ctx* c, const rt_scene_desc* d) {
    if (!d) return fail(c, RT_ERR_INVALID, "scene is NULL");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    if (d->camera.vw <= 0 || d->camera.vh <= 0) return fail(c, RT_ERR_INVALID, "camera: vw and vh must be positive");
    if ((uint64_t)d->camera.vw * (uint64_t)d->camera.vh >= 0x7FFFFFFFull) return fail(c, RT_ERR_UNSUPPORTED, "frame larger than 2^31 pixels");
    if (d->options.max_ray_depth + 2ull > RT_MAX_LEVELS) return fail(c, RT_ERR_UNSUPPORTED, "max_ray_depth above 31 is not supported");
    {
        // every array the description names is dereferenced below: a count without an array is refused here
        const struct { const void* p; uint32_t n; const char* what; } arrays[] = {
            {d->objects, d->n_objects, "objects"}, {d->spheres, d->n_spheres, "spheres"}, {d->planes, d->n_planes, "planes"},
            {d->boxes, d->n_boxes, "boxes"}, {d->meshes, d->n_meshes, "meshes"}, {d->materials, d->n_materials, "materials"},
            {d->textures, d->n_textures, "textures"}, {d->lights, d->n_lights, "lights"}};
        for (const auto& a : arrays)
            if (a.n != 0 && !a.p) return fail(c, RT_ERR_INVALID, std::string("scene: ") + a.what + " is NULL but its count is not 0");
    }
    // the shadow kernels take the light from the grid's y index
    if (d->n_lights > 65535u) return fail(c, RT_ERR_UNSUPPORTED, "more than 65535 lights");
    uint64_t n_records = 0;      // one per sun and point light, one per sample of an area light (one for an area light without samples)
    for (uint32_t i = 0; i < d->n_lights; i++) {
        if (d->lights[i].kind > RT_LIGHT_AREA) return fail(c, RT_ERR_INVALID, "unknown light kind");
        if (d->lights[i].kind == RT_LIGHT_AREA && d->lights[i].area_surface > RT_AREA_RECTANGLE) return fail(c, RT_ERR_INVALID, "unknown area light surface");
        n_records += d->lights[i].kind == RT_LIGHT_AREA ? std::max<uint32_t>(1u, d->lights[i].area_iterations) : 1u;
        if (n_records > 65535u) return fail(c, RT_ERR_UNSUPPORTED, "more than 65535 lights and area light samples");
    }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->materials[i].texture >= d->n_textures) return fail(c, RT_ERR_INVALID, "material: texture index out of range");
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        if (t.kind > RT_TEX_CHECKER) return fail(c, RT_ERR_INVALID, "unknown texture kind");
        if (t.kind == RT_TEX_IMAGE && (!t.rgb || t.w == 0 || t.h == 0)) return fail(c, RT_ERR_INVALID, "image texture without pixels");
    }
    DScene& H = c->hscene;
    std::memset(&H, 0, sizeof(H));
    H.cam = make_camera(d->camera);
    H.vw = d->camera.vw; H.vh = d->camera.vh;
    H.max_depth = d->options.max_ray_depth;
    for (int k = 0; k < 3; k++) H.ambient[k] = (double)d->options.ambient[k] / 255.;   // vector.rs:216-224
    H.n_objects = d->n_objects; H.n_lights = d->n_lights; H.n_meshes = d->n_meshes;

    int rc;
    // textures: one device copy per distinct host pixel pointer
    std::map<const uint8_t*, const uint8_t*> pix;
    std::vector<DMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        const rt_texture& t = d->textures[m.texture];
        DMaterial& dm = mats[i];
        std::memset(&dm, 0, sizeof(dm));
        dm.tex_kind = t.kind; dm.a_rgb = pack_rgb(t.a); dm.b_rgb = pack_rgb(t.b);
        if (t.kind == RT_TEX_IMAGE) {
            dm.tex_w = t.w; dm.tex_h = t.h;
            auto it = pix.find(t.rgb);
            if (it == pix.end()) {
                const uint8_t* dp = nullptr;
                if ((rc = upload(c, t.rgb, (size_t)t.w * t.h * 3, &dp)) != RT_OK) return rc;
                it = pix.emplace(t.rgb, dp).first;
            }
            dm.tex_rgb = it->second;
        }
        dm.reflectiveness = m.reflectiveness; dm.transparency = m.transparency; dm.ior = m.ior; dm.emissivity = m.emissivity;
    }
    if ((rc = upload(c, mats.data(), mats.size(), &H.materials)) != RT_OK) return rc;

    // meshes: one flagged `unchanged` keeps the device copy it had at the same index in the last upload
    std::vector<DMesh> meshes(d->n_meshes);
    {
        std::vector<rt_ctx::MeshCopy> old;
        old.swap(c->mesh_copies);
        c->mesh_copies.resize(d->n_meshes);
        rc = RT_OK;
        for (uint32_t i = 0; i < d->n_meshes && rc == RT_OK; i++) {
            const rt_mesh& m = d->meshes[i];
            rt_ctx::MeshCopy& mc = c->mesh_copies[i];
            if (m.unchanged > 2) rc = fail(c, RT_ERR_INVALID, "mesh " + std::to_string(i) + ": unchanged must be 0, 1 or 2");
            else if (m.unchanged) {
                if (i >= old.size() || old[i].allocs.empty())
                    rc = fail(c, RT_ERR_INVALID, "mesh " + std::to_string(i) + " is flagged unchanged, but the last upload holds no mesh at that index");
                else if (old[i].n_verts != m.n_verts || old[i].n_tris != m.n_tris || (m.unchanged == 1 && (old[i].n_nodes != m.n_nodes || old[i].n_prims != m.n_prims)))
                    rc = fail(c, RT_ERR_INVALID, "mesh " + std::to_string(i) + " is flagged unchanged, but its counts differ from the last upload's");
                else {
                    mc = std::move(old[i]); old[i].allocs.clear(); c->scene_bytes += mc.bytes;
                    if (m.unchanged == 2) rc = refit_mesh(c, m, mc);
                }
            } else {
                const size_t before = c->scene_bytes;
                c->alloc_sink = &mc.allocs;
                rc = upload_mesh(c, m, mc);
                c->alloc_sink = &c->scene_allocs;
                mc.bytes = c->scene_bytes - before;
                mc.n_verts = m.n_verts; mc.n_tris = m.n_tris; mc.n_nodes = m.n_nodes; mc.n_prims = m.n_prims;
            }
            meshes[i] = mc.dm;
        }
        free_mesh_copies(old);
        if (rc != RT_OK) return rc;
    }
    if ((rc = upload(c, meshes.data(), meshes.size(), &H.meshes)) != RT_OK) return rc;

    std::vector<DObject> objs(d->n_objects);
    bool any_far = false, any_refr = false;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        if (o.material >= d->n_materials) return fail(c, RT_ERR_INVALID, "object: material index out of range");
        const uint32_t lim = o.kind == RT_OBJ_SPHERE ? d->n_spheres : o.kind == RT_OBJ_PLANE ? d->n_planes
                           : o.kind == RT_OBJ_BOX ? d->n_boxes : o.kind == RT_OBJ_MESH ? d->n_meshes : 0;
        if (o.kind > RT_OBJ_MESH || o.payload >= lim) return fail(c, RT_ERR_INVALID, "object: bad kind or payload index");
        const rt_material& m = d->materials[o.material];
        objs[i].kind = o.kind; objs[i].material = o.material; objs[i].payload = o.payload;
        objs[i].need_far = (o.kind == RT_OBJ_MESH && m.transparency > RT_EPSILON && m.ior == 1.) ? 1u : 0u;
        if (objs[i].need_far) any_far = true;
        if (o.kind == RT_OBJ_MESH && m.transparency > RT_EPSILON && !(m.ior == 1.)) any_refr = true;   // (NaN ior included)
    }
    // object tree over the spheres and boxes when there are enough of them to pay for it (RTGPU_TLAS_MIN, default 8)
    {
        std::vector<TlasItem> items;
        for (uint32_t i = 0; i < d->n_objects; i++) {
            const rt_object& o = d->objects[i];
            TlasItem it; it.object = i;
            bool finite = true;
            if (o.kind == RT_OBJ_SPHERE) {
                const rt_sphere& sp = d->spheres[o.payload];
                for (int k = 0; k < 3; k++) { it.lo[k] = sp.c[k] - std::fabs(sp.r); it.hi[k] = sp.c[k] + std::fabs(sp.r); }
            } else if (o.kind == RT_OBJ_BOX) {
                const rt_box& b = d->boxes[o.payload];
                for (int k = 0; k < 3; k++) { it.lo[k] = b.pos[k] - std::fabs(b.half[k]); it.hi[k] = b.pos[k] + std::fabs(b.half[k]); }
            } else continue;
            // (finite, and within the range of the half-precision boxes the lane-by-lane walks use)
            for (int k = 0; k < 3; k++) finite = finite && std::isfinite(it.lo[k]) && std::isfinite(it.hi[k]) && std::fabs(it.lo[k]) < 6e4 && std::fabs(it.hi[k]) < 6e4;
            // an object with a NaN / infinite extent stays in the linear scan (and so does one whose index would not fit
            // beside the flag bit of a leaf record)
            // (... or beside the lane number of a queued (object, ray) pair)
            if (finite && o.payload < RT_TLAS_BOX && i < RT_PACKET_MAX_PRIMS) items.push_back(it);
        }
        uint32_t tlas_min = 8;
        if (const char* e = std::getenv("RTGPU_TLAS_MIN")) tlas_min = (uint32_t)std::strtoul(e, nullptr, 10);
        H.tlas = nullptr; H.tlas_quad = nullptr; H.tlas_half = nullptr; H.tlas_geom = nullptr; H.tlas_half_coord_max = 0.f; H.n_tlas = 0;
        std::memset(&H.tlas_mesh, 0, sizeof(H.tlas_mesh));
        if (items.size() >= std::max<uint32_t>(tlas_min, 2)) {
            std::vector<DTlasNode> nodes;
            nodes.reserve(items.size() * 2);
            build_tlas(items, 0, items.size(), nodes);
            for (const TlasItem& it : items) objs[it.object].need_far |= RT_IN_TLAS;
            if ((rc = upload(c, nodes.data(), nodes.size(), &H.tlas)) != RT_OK) return rc;
            H.n_tlas = (uint32_t)nodes.size();
            // the four-wide records: an inner node's two children, each replaced by its own two where it has them
            std::vector<DTlasQuad> quads;
            uint32_t quad_depth = 0;
            auto is_inner = [&](size_t i) { return (nodes[i].object & RT_TLAS_INNER) != 0; };
            std::function<uint32_t(size_t, uint32_t)> fold = [&](size_t i, uint32_t depth) -> uint32_t {
                const uint32_t q = (uint32_t)quads.size();
                quads.emplace_back();
                quad_depth = std::max(quad_depth, depth);
                size_t kids[4]; int nk = 0;
                const size_t pair[2] = {i + 1, (size_t)(nodes[i].object & ~RT_TLAS_INNER)};
                for (size_t ch : pair) {
                    if (is_inner(ch)) { kids[nk++] = ch + 1; kids[nk++] = nodes[ch].object & ~RT_TLAS_INNER; }
                    else kids[nk++] = ch;
                }
                DTlasQuad rec;
                std::memset(&rec, 0, sizeof(rec));
                for (int k = 0; k < 4; k++) rec.ref[k] = RT_TLAS_NONE;
                for (int k = 0; k < nk; k++) {
                    const DTlasNode& nd = nodes[kids[k]];
                    rec.minx[k] = nd.bmin[0]; rec.miny[k] = nd.bmin[1]; rec.minz[k] = nd.bmin[2];
                    rec.maxx[k] = nd.bmax[0]; rec.maxy[k] = nd.bmax[1]; rec.maxz[k] = nd.bmax[2];
                    if (is_inner(kids[k])) rec.ref[k] = fold(kids[k], depth + 1);
                    else {
                        const uint32_t oi = nd.object;
                        rec.ref[k] = RT_TLAS_LEAF | oi;
                        rec.what[k] = (objs[oi].kind == RT_OBJ_BOX ? RT_TLAS_BOX : 0u) | objs[oi].payload;
                    }
                }
                quads[q] = rec;
                return q;
            };
            fold(0, 1);
            // a step of the walk leaves at most three children waiting
            if (3 * (size_t)quad_depth > RT_MAX_STACK) return fail(c, RT_ERR_UNSUPPORTED, "object tree deeper than the traversal stack");
            if ((rc = upload(c, quads.data(), quads.size(), &H.tlas_quad)) != RT_OK) return rc;
            {
                // the same records in 64 bytes: half-precision boxes rounded outward, leaf references that carry the
                // object's index and kind, and the objects' own numbers by object index
                std::vector<DTlasHalf> halves(quads.size());
                std::vector<double> geom((size_t)d->n_objects * 6, 0.);
                float hmax = 0.f;
                for (size_t q = 0; q < quads.size(); q++) {
                    const DTlasQuad& Q = quads[q];
                    DTlasHalf& Hf = halves[q];
                    std::memset(&Hf, 0, sizeof(Hf));
                    for (int k = 0; k < 4; k++) {
                        Hf.ref[k] = Q.ref[k];
                        if (Q.ref[k] == RT_TLAS_NONE) continue;
                        Hf.minx[k] = half_down(Q.minx[k]); Hf.miny[k] = half_down(Q.miny[k]); Hf.minz[k] = half_down(Q.minz[k]);
                        Hf.maxx[k] = half_up(Q.maxx[k]); Hf.maxy[k] = half_up(Q.maxy[k]); Hf.maxz[k] = half_up(Q.maxz[k]);
                        for (uint16_t h : {Hf.minx[k], Hf.miny[k], Hf.minz[k], Hf.maxx[k], Hf.maxy[k], Hf.maxz[k]}) hmax = std::max(hmax, std::fabs(half_value(h)));
                        if (Q.ref[k] & RT_TLAS_LEAF) {
                            const uint32_t oi = Q.ref[k] & ~RT_TLAS_LEAF;
                            const rt_object& o = d->objects[oi];
                            Hf.ref[k] = RT_TLAS_LEAF | (o.kind == RT_OBJ_BOX ? RT_TLAS_BOX : 0u) | oi;
                            if (o.kind == RT_OBJ_SPHERE) { for (int a = 0; a < 3; a++) geom[6 * (size_t)oi + a] = d->spheres[o.payload].c[a]; geom[6 * (size_t)oi + 3] = d->spheres[o.payload].r; }
                            else for (int a = 0; a < 3; a++) { geom[6 * (size_t)oi + a] = d->boxes[o.payload].pos[a]; geom[6 * (size_t)oi + 3 + a] = d->boxes[o.payload].half[a]; }
                        }
                    }
                }
                if ((rc = upload(c, halves.data(), halves.size(), &H.tlas_half)) != RT_OK) return rc;
                if ((rc = upload(c, geom.data(), geom.size(), &H.tlas_geom)) != RT_OK) return rc;
                H.tlas_half_coord_max = hmax;
            }
            // the packet walks take the tree as they take a mesh's: same records, the root's own box, the largest
            // |coordinate| of any box (it bounds the rounding error of their f32 slab arithmetic)
            std::memset(&H.tlas_mesh, 0, sizeof(H.tlas_mesh));
            H.tlas_mesh.nodes = (const DNode*)H.tlas_quad;
            H.tlas_mesh.n_inner = (uint32_t)quads.size();
            H.tlas_mesh.root_ref = 0;
            float cm = 0.f;
            for (const DTlasNode& nd : nodes) for (int k = 0; k < 3; k++) cm = std::max(cm, std::max(std::fabs(nd.bmin[k]), std::fabs(nd.bmax[k])));
            H.tlas_mesh.coord_max = cm;
            for (int k = 0; k < 3; k++) { H.tlas_mesh.root_min[k] = nodes[0].bmin[k]; H.tlas_mesh.root_max[k] = nodes[0].bmax[k]; }
            if (3 * (size_t)quad_depth > RT_PSTACK) return fail(c, RT_ERR_UNSUPPORTED, "object tree deeper than the packet stack");
        }
    }
    if ((rc = upload(c, objs.data(), objs.size(), &H.objects)) != RT_OK) return rc;
    {
        std::vector<uint32_t> linear;
        for (uint32_t i = 0; i < d->n_objects; i++) if (objs[i].kind == RT_OBJ_MESH || !(objs[i].need_far & RT_IN_TLAS)) linear.push_back(i);
        H.n_linear = (uint32_t)linear.size();
        c->rays_need_reciprocal = H.n_tlas != 0;
        for (uint32_t i : linear) if (objs[i].kind == RT_OBJ_BOX) c->rays_need_reciprocal = true;
        if ((rc = upload(c, linear.data(), linear.size(), &H.linear)) != RT_OK) return rc;
    }

    std::vector<double> sph((size_t)d->n_spheres * 4), pln((size_t)d->n_planes * 8), bxs((size_t)d->n_boxes * 6);
    for (uint32_t i = 0; i < d->n_spheres; i++) { for (int k = 0; k < 3; k++) sph[4 * i + k] = d->spheres[i].c[k]; sph[4 * i + 3] = d->spheres[i].r; }
    for (uint32_t i = 0; i < d->n_planes; i++) {
        for (int k = 0; k < 3; k++) { pln[8 * i + k] = d->planes[i].p0[k]; pln[8 * i + 3 + k] = d->planes[i].n[k]; }
        pln[8 * i + 6] = (double)d->planes[i].uv_wrap; pln[8 * i + 7] = 0.;
    }
    for (uint32_t i = 0; i < d->n_boxes; i++) for (int k = 0; k < 3; k++) { bxs[6 * i + k] = d->boxes[i].pos[k]; bxs[6 * i + 3 + k] = d->boxes[i].half[k]; }
    if ((rc = upload(c, sph.data(), sph.size(), &H.spheres)) != RT_OK) return rc;
    if ((rc = upload(c, pln.data(), pln.size(), &H.planes)) != RT_OK) return rc;
    if ((rc = upload(c, bxs.data(), bxs.size(), &H.boxes)) != RT_OK) return rc;

    std::vector<DLight> lights;
    lights.reserve((size_t)n_records);
    bool has_area = false;
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const rt_light& l = d->lights[i];
        DLight dl;
        std::memset(&dl, 0, sizeof(dl));
        dl.kind = l.kind == RT_LIGHT_SUN ? RT_DLIGHT_SUN : RT_DLIGHT_POINT;
        dl.specular_power = l.specular_power; dl.shadows = l.shadows;
        for (int k = 0; k < 3; k++) { dl.col[k] = (double)l.color[k] / 255.; dl.v[k] = l.v[k]; }
        dl.intensity = l.intensity; dl.specular_strength = l.specular_strength;
        dl.shadow_coefficient = l.shadow_coefficient; dl.max_distance = l.max_distance;
        if (l.kind != RT_LIGHT_AREA) { lights.push_back(dl); continue; }
        // an area light: its samples' records in a row (area.rs:117: one point-light evaluation per iteration)
        has_area = true;
        dl.area_surface = l.area_surface; dl.area_n = l.area_iterations; dl.area_li = i; dl.area_radius = l.area_radius;
        for (int q = 0; q < 4; q++) for (int k = 0; k < 3; k++) dl.area_corners[q][k] = l.area_corners[q][k];
        if (l.area_iterations == 0) { dl.kind = RT_DLIGHT_AREA_EMPTY; lights.push_back(dl); continue; }
        dl.kind = RT_DLIGHT_AREA_SAMPLE;
        for (uint32_t it = 0; it < l.area_iterations; it++) { dl.area_it = it; lights.push_back(dl); }
    }
    if (lights.size() != n_records) return fail(c, RT_ERR_INVALID, "light records miscounted");
    H.n_lights = (uint32_t)lights.size();
    H.has_area = has_area ? 1u : 0u; H.area_seed = c->area_seed;
    c->has_area = has_area;
    if ((rc = upload(c, lights.data(), lights.size(), &H.lights)) != RT_OK) return rc;

    H.sky_kind = d->skybox.kind; H.sky_rgb = pack_rgb(d->skybox.color);
    if (d->skybox.kind > RT_SKY_CUBEMAP) return fail(c, RT_ERR_INVALID, "unknown skybox kind");
    if (d->skybox.kind == RT_SKY_CUBEMAP) {
        if (!d->skybox.rgb || d->skybox.w < 4 || d->skybox.w / 4 != d->skybox.h / 3) return fail(c, RT_ERR_INVALID, "cubemap: width/4 must equal height/3 (skybox.rs:49-51)");
        H.sky_w = d->skybox.w; H.sky_h = d->skybox.h;
        if ((rc = upload(c, d->skybox.rgb, (size_t)d->skybox.w * d->skybox.h * 3, &H.sky_pix)) != RT_OK) return rc;
    }
    HIP_TRY(c, hipMemcpy(c->dscene, &H, sizeof(DScene), hipMemcpyHostToDevice));
    c->vw = H.vw; c->vh = H.vh; c->max_depth = H.max_depth; c->n_lights = H.n_lights;
    c->any_need_far = any_far;
    c->any_mesh_refracts = any_refr;
    c->total_tris = 0;
    for (uint32_t i = 0; i < d->n_meshes; i++) c->total_tris += d->meshes[i].n_tris;
    {
        // The level sizes learned from earlier frames say how much room the next frame's levels get.  They carry over to
        // the next frame of a sequence (same objects, meshes kept or refitted), not to another scene that happens to
        // have the same frame size: its first frame would be queued with the old scene's sizes and, outgrowing them,
        // be rendered twice.
        bool same = true;
        for (uint32_t i = 0; i < d->n_meshes; i++) if (d->meshes[i].unchanged == 0) same = false;
        const uint64_t shape = ((uint64_t)d->n_objects * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)H.n_lights << 40) ^ ((uint64_t)d->n_meshes << 20) ^ (c->total_tris << 1) ^ 1ull;
        if (!same || shape != c->scene_shape) { c->learned.clear(); c->whole_frame_rays.clear(); }
        c->scene_shape = shape;
    }
    c->have_scene = true;
    c->ms_upload = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

// The caller's frame buffer is usually fresh from its allocator (scene.rs:220-246 returns a new Vec per frame): pages the
// process has never touched, each of which costs a page fault when the frame's bytes arrive -- at 8K (99.5 MB, 24,000
// pages) more time than the copy over the link.  The faults do not have to wait for the frame: host threads touch the
// buffer's pages (one byte per page; the buffer is an output, whatever it held is overwritten) while the GPU renders.
struct Prefault {
    std::vector<std::thread> th;
    void start(void* dst, size_t bytes, uint32_t n_threads) {
        const size_t PAGE = 4096;
        if (!dst || bytes < ((size_t)8 << 20) || n_threads == 0) return;
        volatile unsigned char* p = static_cast<volatile unsigned char*>(dst);
        const size_t pages = (bytes + PAGE - 1) / PAGE;
        const size_t per = (pages + n_threads - 1) / n_threads;
        for (uint32_t t = 0; t < n_threads; t++) {
            const size_t a = std::min(pages, t * per), b = std::min(pages, a + per);
            if (a < b) th.emplace_back([=]() { for (size_t k = a; k < b; k++) { const size_t off = k * PAGE; p[off < bytes ? off : bytes - 1] = 0; } });
        }
    }
    void join() { for (auto& t : th) t.join(); th.clear(); }
    ~Prefault() { join(); }
};

// A finished frame from device memory into the caller's (pageable) buffer, after whatever is queued on the context's
// stream.  A plain hipMemcpy into pageable memory stages through pinned memory with one host thread behind it (about
// 18 GB/s here); chunked asynchronous copies with an event each depend on the runtime's version (ROCm 7.2: 130 us per
// copy-and-event pair, 3.9 ms for a 4K frame; the runtime PyTorch 2.10 brings along: 0.8 ms).  So the device moves the
// bytes itself: k_copy_out writes the frame into the library's pinned staging buffer, raising one flag per 32 KB, and
// host threads move each block on into the caller's buffer as its flag comes up.  Returns with the bytes in place and
// the stream idle.
static int copy_frame_to_host(rt_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (c->copy_threads == 0 || bytes < ((size_t)4 << 20)) {
        HIP_TRY(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RT_OK;
    }
    const size_t n_blocks = (bytes + RT_COPY_BLOCK - 1) / RT_COPY_BLOCK;
    if (c->h_stage_cap < bytes) {
        if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
        if (c->h_flags) { (void)hipHostFree(c->h_flags); c->h_flags = nullptr; }
        // (coherent: what the device writes is visible to the host after the device's system-scope release)
        if (hipHostMalloc(&c->h_stage, bytes, hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&c->h_flags, n_blocks * sizeof(unsigned int), hipHostMallocCoherent) != hipSuccess)
            return fail(c, RT_ERR_NOMEM, "hipHostMalloc failed for the frame's staging buffer");
        c->h_stage_cap = bytes;
    }
    std::memset(c->h_flags, 0, n_blocks * sizeof(unsigned int));
    std::atomic_thread_fence(std::memory_order_seq_cst);
    char* stage = static_cast<char*>(c->h_stage);
    hipLaunchKernelGGL(k_copy_out, dim3((unsigned)n_blocks), dim3(256), 0, c->stream, static_cast<const uint8_t*>(d_src), reinterpret_cast<uint8_t*>(stage),
                       (unsigned long long)bytes, c->h_flags);
    HIP_TRY(c, hipGetLastError());
    // a mover takes runs of blocks: [k * RUN, (k + 1) * RUN) for k = t, t + n_thr, ...
    const size_t RUN = 8;                                               // 256 KB per turn
    const size_t n_runs = (n_blocks + RUN - 1) / RUN;
    const uint32_t n_thr = (uint32_t)std::min<size_t>(c->copy_threads, n_runs);
    std::atomic<int> bad{0};
    const bool dbg = std::getenv("RTGPU_DEBUG_COPY") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    auto mover = [&](uint32_t t) {
        if (hipSetDevice(c->device) != hipSuccess) { bad = 1; return; }
        for (size_t r = t; r < n_runs && !bad; r += n_thr) {
            const size_t b0 = r * RUN, b1 = std::min(n_blocks, b0 + RUN);
            for (size_t bk = b0; bk < b1; bk++) {
                unsigned long long spins = 0;
                while (__atomic_load_n(&c->h_flags[bk], __ATOMIC_ACQUIRE) == 0u) {
                    if (bad) return;
                    if ((++spins & 0x3FFFull) == 0) {
                        // no flag for a while: has the kernel ended (then its flags are up) or failed?
                        const hipError_t q = hipStreamQuery(c->stream);
                        if (q == hipSuccess) { if (__atomic_load_n(&c->h_flags[bk], __ATOMIC_ACQUIRE) == 0u) { bad = 1; return; } }
                        else if (q != hipErrorNotReady) { bad = 1; return; }
                        std::this_thread::yield();
                    }
                }
            }
            const size_t off = b0 * RT_COPY_BLOCK, len = std::min(bytes, b1 * (size_t)RT_COPY_BLOCK) - off;
            std::memcpy(static_cast<char*>(dst) + off, stage + off, len);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_thr; t++) th.emplace_back(mover, t);
    mover(0);
    for (auto& x : th) x.join();
    const hipError_t es = hipStreamSynchronize(c->stream);
    if (dbg) std::fprintf(stderr, "  copy %zu MB: %.2f ms, threads %u\n", bytes >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(), n_thr);
    if (es != hipSuccess) return fail(c, RT_ERR_DEVICE, std::string("copying the frame to host memory: ") + hipGetErrorString(es));
    if (bad) return fail(c, RT_ERR_DEVICE, "copying the frame to host memory failed");
    return RT_OK;
}

static int seed_one(rt_ctx* c, uint64_t seed);
static int multi_seed(rt_ctx* head, uint64_t seed);
int rt_set_area_seed(rt_ctx* c, uint64_t seed) {
    if (!c) return RT_ERR_INVALID;
    return c->multi ? multi_seed(c, seed) : seed_one(c, seed);
}
static int seed_one(rt_ctx* c, uint64_t seed) {
    c->area_seed = seed;
    if (c->have_scene) {
        HIP_TRY(c, hipSetDevice(c->device));
        c->hscene.area_seed = seed;
        HIP_TRY(c, hipMemcpy(reinterpret_cast<char*>(c->dscene) + offsetof(DScene, area_seed), &seed, sizeof(seed), hipMemcpyHostToDevice));
    }
    return RT_OK;
}

static int multi_render(rt_ctx* head, struct rt_multi& M, uint32_t n_frames, const rt_camera* cameras, uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out);
int rt_render(rt_ctx* c, uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out) {
    if (!c) return RT_ERR_INVALID;
    if (!rgb_out) return fail(c, RT_ERR_INVALID, "rgb_out is NULL");
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (c->multi) return multi_render(c, *c->multi, 1, nullptr, rgb_out, prim_id_out, stats_out);
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t px = (size_t)c->vw * (size_t)c->vh;
    Prefault touch, touch_ids;
    touch.start(rgb_out, px * 3, c->copy_threads);
    touch_ids.start(prim_id_out, px * 8, c->copy_threads);
    const int64_t slots = rt_shard_pixels(c->vw, c->vh, 0, 1);
    int rc;
    if ((rc = ensure_buf(c, &c->d_out, &c->d_out_cap, px * 3)) != RT_OK) return rc;
    if (prim_id_out) {
        if ((rc = ensure_buf(c, &c->d_ids, &c->d_ids_cap, (size_t)slots * 8)) != RT_OK) return rc;
        if ((rc = ensure_buf(c, &c->d_ids2, &c->d_ids2_cap, px * 8)) != RT_OK) return rc;
    }
    rt_stats st;
    rc = render_frame(c, 0, 1, 1, static_cast<uint8_t*>(c->d_out), prim_id_out ? static_cast<uint32_t*>(c->d_ids) : nullptr, &st);
    if (rc != RT_OK) return rc;
    if (prim_id_out) {
        ShardMap map = make_map(c->vw, c->vh, 0, 1, 1);
        hipLaunchKernelGGL(k_ids_rowmajor, dim3(((uint32_t)slots + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->dscene,
                           static_cast<const uint32_t*>(c->d_ids), static_cast<uint32_t*>(c->d_ids2), (uint32_t)slots, map);
        touch_ids.join();      // (before any byte of the frame lands: the threads that touch the pages write to them)
        if ((rc = copy_frame_to_host(c, prim_id_out, c->d_ids2, px * 8)) != RT_OK) return rc;
    }
    touch.join();
    if ((rc = copy_frame_to_host(c, rgb_out, c->d_out, px * 3)) != RT_OK) return rc;
    HIP_TRY(c, hipGetLastError());
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st.device_bytes = c->scene_bytes + c->work_bytes;
    if (stats_out) *stats_out = st;
    return RT_OK;
}

int rt_render_shard(rt_ctx* c, uint32_t shard, uint32_t n_shards, void* d_rgb, void* d_ids, rt_stats* stats_out) {
    if (!c) return RT_ERR_INVALID;
    if (!d_rgb) return fail(c, RT_ERR_INVALID, "d_rgb is NULL");
    if (n_shards == 0 || shard >= n_shards) return fail(c, RT_ERR_INVALID, "bad shard arguments");
    auto t0 = std::chrono::steady_clock::now();
    rt_stats st;
    int rc = render_frame(c, shard, n_shards, 0, static_cast<uint8_t*>(d_rgb), static_cast<uint32_t*>(d_ids), &st);
    if (rc != RT_OK) return rc;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) *stats_out = st;
    return RT_OK;
}

int rt_render_shard_batch(rt_ctx* c, uint32_t shard, uint32_t n_shards, uint32_t n_frames, const rt_camera* cameras,
                          int64_t frame_stride, void* d_rgb, void* d_ids, rt_stats* stats_out) {
    if (!c) return RT_ERR_INVALID;
    if (!d_rgb) return fail(c, RT_ERR_INVALID, "d_rgb is NULL");
    if (n_shards == 0 || shard >= n_shards) return fail(c, RT_ERR_INVALID, "bad shard arguments");
    if (n_frames == 0 || n_frames > RT_BATCH_MAX) return fail(c, RT_ERR_INVALID, "rt_render_shard_batch: n_frames must be 1.." + std::to_string(RT_BATCH_MAX));
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (frame_stride < rt_shard_pixels(c->vw, c->vh, shard, n_shards) || frame_stride >= 0x7FFFFFFFll)
        return fail(c, RT_ERR_INVALID, "rt_render_shard_batch: frame_stride smaller than the shard (rt_shard_pixels)");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if (cameras) {
        c->h_cams.resize(n_frames);
        for (uint32_t f = 0; f < n_frames; f++) {
            if (cameras[f].vw != c->vw || cameras[f].vh != c->vh)
                return fail(c, RT_ERR_INVALID, "rt_render_shard_batch: camera " + std::to_string(f) + " has another frame size than the uploaded scene");
            c->h_cams[f] = make_camera(cameras[f]);
        }
        if ((rc = ensure_buf(c, &c->d_cams, &c->d_cams_cap, RT_BATCH_MAX * sizeof(DCamera))) != RT_OK) return rc;
        // (pageable host memory: the copy has left h_cams when the call returns)
        HIP_TRY(c, hipMemcpyAsync(c->d_cams, c->h_cams.data(), n_frames * sizeof(DCamera), hipMemcpyHostToDevice, c->stream));
    }
    rt_stats st;
    rc = render_frame(c, shard, n_shards, 0, static_cast<uint8_t*>(d_rgb), static_cast<uint32_t*>(d_ids), &st, n_frames,
                      cameras ? static_cast<const DCamera*>(c->d_cams) : nullptr, (uint32_t)frame_stride);
    if (rc != RT_OK) return rc;
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) *stats_out = st;
    return RT_OK;
}

int rt_detile_frames(rt_ctx* c, const void* d_gathered, uint32_t n_shards, uint32_t n_frames, int64_t slots_per_shard, uint32_t elem_bytes, void* d_images) {
    if (!c) return RT_ERR_INVALID;
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (!d_gathered || !d_images || n_shards == 0 || n_frames == 0 || slots_per_shard <= 0 || (elem_bytes != 3 && elem_bytes != 8)) return fail(c, RT_ERR_INVALID, "rt_detile: bad arguments");
    for (uint32_t s = 0; s < n_shards; s++)
        if (rt_shard_pixels(c->vw, c->vh, s, n_shards) > slots_per_shard) return fail(c, RT_ERR_INVALID, "rt_detile: slots_per_shard smaller than a shard");
    HIP_TRY(c, hipSetDevice(c->device));
    ShardMap map = make_map(c->vw, c->vh, 0, n_shards, 1);
    const unsigned long long total = (unsigned long long)slots_per_shard * n_shards * n_frames;
    if (total >= (1ull << 40)) return fail(c, RT_ERR_UNSUPPORTED, "rt_detile: too many slots");
    hipLaunchKernelGGL(k_detile, dim3((unsigned)((total + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, c->stream, static_cast<const uint8_t*>(d_gathered),
                       static_cast<uint8_t*>(d_images), c->vw, c->vh, n_shards, n_frames, (unsigned long long)slots_per_shard, elem_bytes, map);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int rt_detile(rt_ctx* c, const void* d_gathered, uint32_t n_shards, int64_t slots_per_shard, uint32_t elem_bytes, void* d_image) {
    return rt_detile_frames(c, d_gathered, n_shards, 1, slots_per_shard, elem_bytes, d_image);
}

int rt_cast_rays(rt_ctx* c, uint32_t n, const double* origins, const double* dirs, rt_ray_hit* hits_out) {
    if (!c) return RT_ERR_INVALID;
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!origins || !dirs || !hits_out) return fail(c, RT_ERR_INVALID, "rt_cast_rays: NULL argument");
    static_assert(sizeof(DProbeHit) == sizeof(rt_ray_hit), "probe record layout");
    HIP_TRY(c, hipSetDevice(c->device));
    double *d_o = nullptr, *d_d = nullptr; DProbeHit* d_h = nullptr;
    int rc = RT_OK;
    if (hipMalloc((void**)&d_o, (size_t)n * 24) != hipSuccess || hipMalloc((void**)&d_d, (size_t)n * 24) != hipSuccess ||
        hipMalloc((void**)&d_h, (size_t)n * sizeof(DProbeHit)) != hipSuccess) {
        rc = fail(c, RT_ERR_NOMEM, "rt_cast_rays: hipMalloc failed");
    } else {
        hipError_t e = hipMemcpy(d_o, origins, (size_t)n * 24, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_d, dirs, (size_t)n * 24, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_probe, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->dscene, d_o, d_d, d_h, n);
            e = hipStreamSynchronize(c->stream);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hits_out, d_h, (size_t)n * sizeof(DProbeHit), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, RT_ERR_DEVICE, std::string("rt_cast_rays: ") + hipGetErrorString(e));
    }
    if (d_o) (void)hipFree(d_o);
    if (d_d) (void)hipFree(d_d);
    if (d_h) (void)hipFree(d_h);
    return rc;
}

// rt_cast_rays without any allocation: the rays and their hit records are staged in this device's copy of the
// file-scope array.  All calls on one GPU share that array.
int rt_cast_rays_noalloc(rt_ctx* c, uint32_t n, const double* origins, const double* dirs, rt_ray_hit* hits_out) {
    if (!c) return RT_ERR_INVALID;
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!origins || !dirs || !hits_out) return fail(c, RT_ERR_INVALID, "rt_cast_rays_noalloc: NULL argument");
    if (n > RT_CAST_NOALLOC_MAX)
        return fail(c, RT_ERR_INVALID, "rt_cast_rays_noalloc: " + std::to_string(n) + " rays exceed the limit of " + std::to_string(RT_CAST_NOALLOC_MAX) +
                                           " per call (RT_CAST_NOALLOC_MAX); split the batch or use rt_cast_rays");
    if (!c->scratch) return fail(c, RT_ERR_DEVICE, "rt_cast_rays_noalloc: the staging array's address is unknown");
    static_assert(sizeof(DProbeHit) == sizeof(rt_ray_hit), "probe record layout");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned char* base = static_cast<unsigned char*>(c->scratch);
    double* d_o = reinterpret_cast<double*>(base + RT_NOALLOC_ORIGINS_OFF);
    double* d_d = reinterpret_cast<double*>(base + RT_NOALLOC_DIRS_OFF);
    DProbeHit* d_h = reinterpret_cast<DProbeHit*>(base + RT_NOALLOC_HITS_OFF);
    // nothing in the array has to start at zero: the first n rays are overwritten here and the kernel writes every
    // field of the first n hit records, which are the only ones copied out
    HIP_TRY(c, hipMemcpyAsync(d_o, origins, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_d, dirs, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_probe, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->stream, c->dscene, d_o, d_d, d_h, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(hits_out, d_h, (size_t)n * sizeof(DProbeHit), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ---- One context over several GPUs of a node (rt_ctx_create_multi) ---------------------------------------------------
//
// The reference's callers are one process that asks for one frame (sdl/src/main.rs:61,91 -> scene.rs:249-250 ->
// scene.rs:220).  For that caller several GPUs must look like one context: the scene goes to every device at upload,
// device k renders shard k of n (the image-space shards of rt_render_shard, pixels being independent, scene.rs:239-242),
// the compact shard buffers are gathered to the first device -- RCCL's ncclGather over the devices' direct links, one
// communicator per device inside this process -- and the first device puts the frame together (k_detile) and copies
// it out.  One host thread per device drives its shard; there is no exchange between devices other than that gather.
//
// RCCL is opened with dlopen when the first such context is created: a process that renders on one GPU never loads it.
// A device list that names a GPU twice (two shards on one GPU: a rehearsal of the n > 1 path on a one-GPU machine, which
// RCCL refuses) is gathered with device-to-device copies instead; RTGPU_MULTI_GATHER=copy asks for that outright.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static std::mutex g_rccl_mutex;
static RcclApi g_rccl;
static const RcclApi* rccl_api(std::string& err) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { const char* e = dlerror(); err = std::string("librccl could not be opened: ") + (e ? e : "?"); return nullptr; }
    RcclApi a;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.Gather = reinterpret_cast<decltype(a.Gather)>(dlsym(h, "ncclGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Gather || !a.GetErrorString) {
        err = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclGather / ncclGetErrorString";
        dlclose(h);
        return nullptr;
    }
    a.handle = h;
    g_rccl = a;
    return &g_rccl;
}

struct rt_multi {
    std::vector<rt_ctx*> member;      // member[0] is the context the caller holds; member[k] renders shard k
    std::vector<int> device;
    const RcclApi* api = nullptr;     // NULL: gathered with device-to-device copies
    std::vector<ncclComm_t> comm;
    // a member's compact shard buffers, on its device (the first member's are the head of the gathered buffers)
    struct Part { void* rgb = nullptr; size_t rgb_cap = 0; void* ids = nullptr; size_t ids_cap = 0; int rc = RT_OK; rt_stats st; };
    std::vector<Part> part;
    void* gath_rgb = nullptr; size_t gath_rgb_cap = 0;     // on the first device: [shard][frame][slot]
    void* gath_ids = nullptr; size_t gath_ids_cap = 0;
    void* frames = nullptr; size_t frames_cap = 0;         // on the first device: the row-major frames of a set
};

int rt_multi_layout(int32_t vw, int32_t vh, uint32_t n_devices, uint32_t n_frames, rt_multi_plan* out) {
    if (!out || vw <= 0 || vh <= 0 || n_devices == 0 || n_frames == 0 || n_frames > RT_BATCH_MAX) return RT_ERR_INVALID;
    // every device contributes the same count to the gather: its shard padded to the largest one, shard 0's
    const int64_t slots = rt_shard_pixels(vw, vh, 0, n_devices);
    if (slots <= 0 || (uint64_t)slots * n_frames >= 0x7FFFFFFFull) return RT_ERR_UNSUPPORTED;
    out->slots_per_shard = slots;
    out->part_bytes_rgb = slots * 3 * (int64_t)n_frames;
    out->part_bytes_ids = slots * 8 * (int64_t)n_frames;
    out->gathered_bytes_rgb = out->part_bytes_rgb * (int64_t)n_devices;
    out->gathered_bytes_ids = out->part_bytes_ids * (int64_t)n_devices;
    out->frame_bytes_rgb = (int64_t)vw * vh * 3;
    return RT_OK;
}

int rt_ctx_device_count(const rt_ctx* c) { return !c ? 0 : c->multi ? (int)c->multi->member.size() : 1; }

int rt_ctx_create_multi(const int* device_ids, int n_devices, rt_ctx** out) {
    if (!out) { g_create_error = "rt_ctx_create_multi: out is NULL"; return RT_ERR_INVALID; }
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > RT_MAX_DEVICES) { g_create_error = "rt_ctx_create_multi: bad device list"; return RT_ERR_INVALID; }
    // one device: the context of rt_ctx_create, and no collective anywhere (RTGPU_MULTI_FORCE=1 keeps the gather
    // machinery even then, so that a one-GPU machine can run the RCCL calls: a communicator of one rank)
    const char* force = std::getenv("RTGPU_MULTI_FORCE");
    if (n_devices == 1 && !(force && force[0] == '1')) return rt_ctx_create(device_ids[0], out);
    std::unique_ptr<rt_multi> M(new (std::nothrow) rt_multi());
    if (!M) { g_create_error = "out of host memory"; return RT_ERR_NOMEM; }
    auto undo = [&]() { for (rt_ctx* m : M->member) rt_ctx_destroy(m); M->member.clear(); };
    for (int k = 0; k < n_devices; k++) {
        rt_ctx* m = nullptr;
        const int rc = rt_ctx_create(device_ids[k], &m);
        if (rc != RT_OK) { g_create_error = "device " + std::to_string(device_ids[k]) + ": " + g_create_error; undo(); return rc; }
        M->member.push_back(m);
        M->device.push_back(device_ids[k]);
    }
    M->part.resize((size_t)n_devices);
    bool repeated = false;
    for (int a = 0; a < n_devices; a++) for (int b = a + 1; b < n_devices; b++) if (device_ids[a] == device_ids[b]) repeated = true;
    const char* how = std::getenv("RTGPU_MULTI_GATHER");
    const bool by_copy = repeated || (how && std::string(how) == "copy");
    if (!by_copy) {
        std::string err;
        M->api = rccl_api(err);
        if (!M->api) { g_create_error = err; undo(); return RT_ERR_DEVICE; }
        M->comm.assign((size_t)n_devices, nullptr);
        const ncclResult_t r = M->api->CommInitAll(M->comm.data(), n_devices, device_ids);
        if (r != ncclSuccess) {
            g_create_error = std::string("ncclCommInitAll: ") + M->api->GetErrorString(r);
            M->comm.clear(); undo();
            return RT_ERR_DEVICE;
        }
    }
    rt_ctx* head = M->member[0];
    head->multi = M.release();
    (void)hipSetDevice(head->device);
    *out = head;
    return RT_OK;
}

static void multi_destroy(rt_ctx* head) {
    if (head->solo) {
        rt_multi* S = head->solo;
        head->solo = nullptr;
        (void)hipSetDevice(head->device);
        if (head->stream) (void)hipStreamSynchronize(head->stream);
        if (S->gath_rgb) (void)hipFree(S->gath_rgb);
        if (S->gath_ids) (void)hipFree(S->gath_ids);
        if (S->frames) (void)hipFree(S->frames);
        delete S;
    }
    rt_multi* M = head->multi;
    if (!M) return;
    head->multi = nullptr;
    for (size_t k = 0; k < M->member.size(); k++) {
        rt_ctx* m = M->member[k];
        (void)hipSetDevice(m->device);
        if (m->stream) (void)hipStreamSynchronize(m->stream);
        if (k > 0) {
            if (M->part[k].rgb) (void)hipFree(M->part[k].rgb);
            if (M->part[k].ids) (void)hipFree(M->part[k].ids);
        }
    }
    if (M->api) for (ncclComm_t cm : M->comm) if (cm) (void)M->api->CommDestroy(cm);
    (void)hipSetDevice(head->device);
    if (M->gath_rgb) (void)hipFree(M->gath_rgb);
    if (M->gath_ids) (void)hipFree(M->gath_ids);
    if (M->frames) (void)hipFree(M->frames);
    for (size_t k = 1; k < M->member.size(); k++) rt_ctx_destroy(M->member[k]);
    delete M;
}

// fn(k) for every member, each on a host thread of its own (the first on the calling thread); the first failure's code,
// its message filed in the head context with the device it came from
extern "C++" {
template <typename F>
static int multi_each(rt_ctx* head, rt_multi& M, F fn) {
    const size_t n = M.member.size();
    std::vector<int> rc(n, RT_OK);
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; k++) th.emplace_back([&, k]() { rc[k] = fn(k); });
    rc[0] = fn(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(head->device);
    for (size_t k = 0; k < n; k++)
        if (rc[k] != RT_OK) {
            if (k > 0) head->err = "device " + std::to_string(M.device[k]) + " (shard " + std::to_string(k) + "): " + M.member[k]->err;
            return rc[k];
        }
    return RT_OK;
}
}  // extern "C++"

static int multi_upload(rt_ctx* head, const rt_scene_desc* d) {
    rt_multi& M = *head->multi;
    // every device gets its own deep copy of the scene (a few hundred megabytes at most against 288 GB per device),
    // re-emitted by that member's own host thread; rt_mesh::unchanged means the same on every member
    const int rc = multi_each(head, M, [&](size_t k) { return upload_one(M.member[k], d); });
    if (rc != RT_OK) {   // a failed upload leaves nothing behind on any device
        const std::string keep = head->err;
        for (rt_ctx* m : M.member) { (void)hipSetDevice(m->device); free_scene(m); free_mesh_copies(m->mesh_copies); }
        (void)hipSetDevice(head->device);
        head->err = keep;
    }
    return rc;
}

static int multi_seed(rt_ctx* head, uint64_t seed) {
    rt_multi& M = *head->multi;
    for (size_t k = 0; k < M.member.size(); k++) {
        const int rc = seed_one(M.member[k], seed);
        if (rc != RT_OK) { if (k > 0) head->err = M.member[k]->err; (void)hipSetDevice(head->device); return rc; }
    }
    (void)hipSetDevice(head->device);
    return RT_OK;
}

// the members' compact buffers -> the first device, shard-major
static int multi_gather(rt_ctx* head, rt_multi& M, bool ids, size_t bytes_per_member) {
    const size_t n = M.member.size();
    char* gath = static_cast<char*>(ids ? M.gath_ids : M.gath_rgb);
    auto src = [&](size_t k) { return static_cast<const char*>(k == 0 ? (const void*)gath : (ids ? M.part[k].ids : M.part[k].rgb)); };
    if (M.api) {
        // one ncclGather per device, issued from this one thread as a group; the first member's own part already lies
        // where the gather would put it (in place: sendbuff == recvbuff + rank * count)
        ncclResult_t r = M.api->GroupStart();
        for (size_t k = 0; k < n && r == ncclSuccess; k++) {
            (void)hipSetDevice(M.device[k]);
            r = M.api->Gather(src(k), k == 0 ? gath : nullptr, bytes_per_member, ncclUint8, 0, M.comm[k], M.member[k]->stream);
        }
        const ncclResult_t r2 = M.api->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { (void)hipSetDevice(head->device); return fail(head, RT_ERR_DEVICE, std::string("ncclGather: ") + M.api->GetErrorString(r)); }
        // the senders' buffers are free again, and the gathered bytes in place, once every stream has run its part
        for (size_t k = n; k-- > 0;) {
            (void)hipSetDevice(M.device[k]);
            const hipError_t e = hipStreamSynchronize(M.member[k]->stream);
            if (e != hipSuccess) { (void)hipSetDevice(head->device); return fail(head, RT_ERR_DEVICE, std::string("gather: ") + hipGetErrorString(e)); }
        }
    } else {
        (void)hipSetDevice(head->device);
        for (size_t k = 1; k < n; k++)
            HIP_TRY(head, hipMemcpyPeerAsync(gath + k * bytes_per_member, head->device, src(k), M.device[k], bytes_per_member, head->stream));
        HIP_TRY(head, hipStreamSynchronize(head->stream));
    }
    (void)hipSetDevice(head->device);
    return RT_OK;
}

// n_frames frames (cameras == NULL: the uploaded scene's own view) over all members: render, gather, put together, copy out
static int multi_render(rt_ctx* head, rt_multi& M, uint32_t n_frames, const rt_camera* cameras, uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out) {
    const uint32_t n = (uint32_t)M.member.size();
    auto t0 = std::chrono::steady_clock::now();
    rt_multi_plan plan;
    {
        const int prc = rt_multi_layout(head->vw, head->vh, n, n_frames, &plan);
        if (prc != RT_OK) return fail(head, prc, "frame (or set of frames) too large for one set of launches");
    }
    HIP_TRY(head, hipSetDevice(head->device));
    int rc;
    Prefault touch, touch_ids;
    touch.start(rgb_out, (size_t)plan.frame_bytes_rgb * n_frames, head->copy_threads);
    touch_ids.start(prim_id_out, (size_t)head->vw * head->vh * 8 * n_frames, head->copy_threads);
    if ((rc = ensure_buf(head, &M.gath_rgb, &M.gath_rgb_cap, (size_t)plan.gathered_bytes_rgb)) != RT_OK) return rc;
    if (prim_id_out && (rc = ensure_buf(head, &M.gath_ids, &M.gath_ids_cap, (size_t)plan.gathered_bytes_ids)) != RT_OK) return rc;
    if ((rc = ensure_buf(head, &M.frames, &M.frames_cap, (size_t)plan.frame_bytes_rgb * n_frames)) != RT_OK) return rc;
    if (prim_id_out && (rc = ensure_buf(head, &head->d_ids2, &head->d_ids2_cap, (size_t)head->vw * head->vh * 8 * n_frames)) != RT_OK) return rc;
    rc = multi_each(head, M, [&](size_t k) -> int {
        rt_ctx* m = M.member[k];
        rt_multi::Part& P = M.part[k];
        if (hipSetDevice(m->device) != hipSuccess) return fail(m, RT_ERR_DEVICE, "hipSetDevice failed");
        void *rgb = M.gath_rgb, *ids = prim_id_out ? M.gath_ids : nullptr;
        if (k > 0) {
            int r;
            if ((r = ensure_buf(m, &P.rgb, &P.rgb_cap, (size_t)plan.part_bytes_rgb)) != RT_OK) return r;
            if (prim_id_out && (r = ensure_buf(m, &P.ids, &P.ids_cap, (size_t)plan.part_bytes_ids)) != RT_OK) return r;
            rgb = P.rgb; ids = prim_id_out ? P.ids : nullptr;
        }
        if (n_frames == 1 && !cameras) return rt_render_shard(m, (uint32_t)k, n, rgb, ids, &P.st);
        return rt_render_shard_batch(m, (uint32_t)k, n, n_frames, cameras, plan.slots_per_shard, rgb, ids, &P.st);
    });
    if (rc != RT_OK) return rc;
    const bool dbg = std::getenv("RTGPU_DEBUG_MULTI") != nullptr;
    auto lap = [&](const char* what) {
        if (dbg) std::fprintf(stderr, "  multi_render %-8s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    lap("rendered");
    if (n > 1 || M.api) {
        if ((rc = multi_gather(head, M, false, (size_t)plan.part_bytes_rgb)) != RT_OK) return rc;
        if (prim_id_out && (rc = multi_gather(head, M, true, (size_t)plan.part_bytes_ids)) != RT_OK) return rc;
    }
    if ((rc = rt_detile_frames(head, M.gath_rgb, n, n_frames, plan.slots_per_shard, 3, M.frames)) != RT_OK) return rc;
    if (prim_id_out) {
        if ((rc = rt_detile_frames(head, M.gath_ids, n, n_frames, plan.slots_per_shard, 8, head->d_ids2)) != RT_OK) return rc;
        touch_ids.join();
        if ((rc = copy_frame_to_host(head, prim_id_out, head->d_ids2, (size_t)head->vw * head->vh * 8 * n_frames)) != RT_OK) return rc;
    }
    lap("detiled");
    touch.join();
    lap("touched");
    if ((rc = copy_frame_to_host(head, rgb_out, M.frames, (size_t)plan.frame_bytes_rgb * n_frames)) != RT_OK) return rc;
    lap("copied");
    HIP_TRY(head, hipGetLastError());
    if (stats_out) {
        // rays add up over the shards; times are the slowest member's (the members work side by side)
        rt_stats st = M.part[0].st;
        for (uint32_t k = 1; k < n; k++) {
            const rt_stats& p = M.part[k].st;
            st.rays_primary += p.rays_primary; st.rays_reflect += p.rays_reflect; st.rays_transmit += p.rays_transmit; st.rays_shadow += p.rays_shadow;
            for (int d = 0; d < 8; d++) st.rays_per_level[d] += p.rays_per_level[d];
            st.ms_render = std::max(st.ms_render, p.ms_render); st.ms_trace = std::max(st.ms_trace, p.ms_trace);
            st.ms_shadow = std::max(st.ms_shadow, p.ms_shadow); st.ms_shade = std::max(st.ms_shade, p.ms_shade);
            st.ms_resolve = std::max(st.ms_resolve, p.ms_resolve); st.ms_upload = std::max(st.ms_upload, p.ms_upload);
            st.ms_shadow_main = std::max(st.ms_shadow_main, p.ms_shadow_main);
            st.device_bytes += p.device_bytes;
            st.nodes_visited += p.nodes_visited; st.tris_tested += p.tris_tested; st.wave_nodes += p.wave_nodes; st.wave_tris += p.wave_tris;
            st.long_trace += p.long_trace; st.long_shadow += p.long_shadow; st.bytes_fetched += p.bytes_fetched; st.bytes_fetched_main += p.bytes_fetched_main;
            st.max_lane_steps = std::max(st.max_lane_steps, p.max_lane_steps); st.max_wave_steps = std::max(st.max_wave_steps, p.max_wave_steps);
            st.speculative = st.speculative && p.speculative ? 1u : 0u;
            st.respeculated += p.respeculated;
        }
        st.device_bytes += M.gath_rgb_cap + M.gath_ids_cap + M.frames_cap;
        st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *stats_out = st;
    }
    return RT_OK;
}

int rt_render_frames(rt_ctx* c, uint32_t n_frames, const rt_camera* cameras, uint8_t* rgb_out, uint32_t* prim_id_out, rt_stats* stats_out) {
    if (!c) return RT_ERR_INVALID;
    if (!rgb_out) return fail(c, RT_ERR_INVALID, "rgb_out is NULL");
    if (!c->have_scene) return fail(c, RT_ERR_INVALID, "no scene uploaded");
    if (n_frames == 0 || n_frames > RT_BATCH_MAX) return fail(c, RT_ERR_INVALID, "rt_render_frames: n_frames must be 1.." + std::to_string(RT_BATCH_MAX));
    if (cameras)
        for (uint32_t f = 0; f < n_frames; f++)
            if (cameras[f].vw != c->vw || cameras[f].vh != c->vh)
                return fail(c, RT_ERR_INVALID, "rt_render_frames: camera " + std::to_string(f) + " has another frame size than the uploaded scene");
    if (!c->multi) {
        // a context on one device renders its sets of frames the same way: one member, no gather.  Its buffers are kept
        // beside the context for the next set (rt_ctx_destroy frees them); the other entry points do not look at them.
        if (!c->solo) {
            c->solo = new (std::nothrow) rt_multi();
            if (!c->solo) return fail(c, RT_ERR_NOMEM, "out of host memory");
            c->solo->member.push_back(c); c->solo->device.push_back(c->device); c->solo->part.resize(1);
        }
        return multi_render(c, *c->solo, n_frames, cameras, rgb_out, prim_id_out, stats_out);
    }
    return multi_render(c, *c->multi, n_frames, cameras, rgb_out, prim_id_out, stats_out);
}

}  // extern "C"
