// Device-side construction (linear BVH over a Morton curve) and refitting of a mesh's four-wide BVH: see bvh_build.h.
//
// Build: triangle centroids (Aabb::centroid of the reference's Triangle primitive, acceleration.rs:457-468: the
// middle of the triangle's box) -> 63-bit Morton keys -> radix sort -> binary radix tree over the sorted keys, all
// inner nodes at once (Karras 2012) -> every other level of that tree becomes one four-wide record (a node adopts its
// grandchildren), numbered by a prefix sum -> triangle records in leaf order (one triangle per leaf) -> boxes by the
// refit pass.  Refit: every record's thread works out the boxes of its leaf children, then climbs: the last of a
// record's children to arrive (an atomic counter per record) forms the record's own box, writes it into the
// parent's slot and goes on up.
#include "bvh_build.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace {

#define BB 256
#define GRID(n) dim3((unsigned)(((n) + BB - 1) / BB))
typedef unsigned long long u64;

// order-preserving map of f32 onto u32, for atomicMin / atomicMax
__device__ inline unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f; memcpy(&f, &v, 4); return f;
}

__device__ inline void tri_box(const double* verts, const uint32_t* tri_pos, uint32_t t, double mn[3], double mx[3]) {
    const double* a = verts + 3 * (size_t)tri_pos[3 * (size_t)t];
    const double* b = verts + 3 * (size_t)tri_pos[3 * (size_t)t + 1];
    const double* c = verts + 3 * (size_t)tri_pos[3 * (size_t)t + 2];
    for (int k = 0; k < 3; k++) { mn[k] = fmin(fmin(a[k], b[k]), c[k]); mx[k] = fmax(fmax(a[k], b[k]), c[k]); }
}

// bounds of the triangles' centroids: cb[0..2] = min, cb[3..5] = max (ordered encoding)
__global__ __launch_bounds__(BB) void k_centroid_bounds(const double* verts, const uint32_t* tri_pos, uint32_t n, unsigned* cb) {
    const uint32_t t = blockIdx.x * BB + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (t < n) {
        double mn[3], mx[3];
        tri_box(verts, tri_pos, t, mn, mx);
        for (int k = 0; k < 3; k++) { const float c = (float)((mn[k] + mx[k]) * 0.5); lo[k] = c; hi[k] = c; }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off)); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) {
            if (lo[k] <= hi[k]) { atomicMin(&cb[k], f2ord(lo[k])); atomicMax(&cb[3 + k], f2ord(hi[k])); }
        }
}

// Morton keys whose 63 bits are dealt to the axes by extent: each bit halves the currently longest side of the cell,
// so a flat mesh (a terrain: two long axes, one short) spends its bits where they separate triangles instead of
// cycling x, y, z regardless (which splits along the short axis again and again and makes overlapping boxes).
__global__ __launch_bounds__(BB) void k_morton(const double* verts, const uint32_t* tri_pos, uint32_t n, const unsigned* cb, u64* keys, uint32_t* idx) {
    const uint32_t t = blockIdx.x * BB + threadIdx.x;
    if (t >= n) return;
    double mn[3], mx[3];
    tri_box(verts, tri_pos, t, mn, mx);
    double lo[3], ext[3], f[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = (double)ord2f(cb[k]);
        const double hi = (double)ord2f(cb[3 + k]);
        ext[k] = hi > lo[k] ? hi - lo[k] : 0.;
        const double c = (mn[k] + mx[k]) * 0.5;
        double v = ext[k] > 0. ? (c - lo[k]) / ext[k] : 0.;
        v = v < 0. ? 0. : v > 1. ? 1. : v;
        f[k] = (v == v) ? v : 0.;               // (NaN coordinates fall to 0)
    }
    // how many of the 63 bits each axis gets: every bit halves the currently longest side (same in every thread)
    int bits[3] = {0, 0, 0};
    double e[3] = {ext[0], ext[1], ext[2]};
    for (int b = 0; b < 63; b++) {
        int a = 0;
        if (e[1] > e[a]) a = 1;
        if (e[2] > e[a]) a = 2;
        e[a] *= 0.5; bits[a]++;
    }
    // quantised coordinates, bits[a] bits each
    u64 q[3];
    // (beyond 53 bits the constant rounds up to 2^bits as a double and f = 1 would land one past the last cell: clamped)
    for (int k = 0; k < 3; k++) {
        const u64 top = bits[k] > 0 ? (1ull << bits[k]) - 1ull : 0ull;
        const u64 v = bits[k] > 0 ? (u64)(f[k] * (double)top) : 0ull;
        q[k] = v > top ? top : v;
    }
    // the bits in that same order, most significant first
    u64 key = 0;
    int left[3] = {bits[0], bits[1], bits[2]};
    double e2[3] = {ext[0], ext[1], ext[2]};
    for (int b = 0; b < 63; b++) {
        int a = 0;
        if (e2[1] > e2[a]) a = 1;
        if (e2[2] > e2[a]) a = 2;
        e2[a] *= 0.5;
        left[a]--;
        key = (key << 1) | ((q[a] >> left[a]) & 1ull);
    }
    keys[t] = key;
    idx[t] = t;
}

// common prefix of the keys at sorted positions i and j (equal keys are told apart by their positions), -1 off the ends
__device__ inline int prefix(const u64* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const u64 a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// A subtree of the radix tree that covers at most this many triangles becomes one leaf (its triangles are neighbours on
// the curve and lie in one run of the sorted order).  The reference's own builder stops splitting where a split does
// not pay (acceleration.rs:356) and ends with two triangles per leaf on a regular mesh; one triangle per leaf means
// twice the inner nodes for the walks to step through.
#ifndef RT_LBVH_LEAF
#define RT_LBVH_LEAF 2
#endif

// One inner node per thread (n - 1 of them over n sorted keys): the range of keys it covers and where it splits.
// child[2 i], child[2 i + 1]: RT_LEAF_BIT | first sorted position of a leaf, or an inner node's index.  Node 0 is the
// root.  covers[i]: how many keys node i covers; a node that covers at most RT_LBVH_LEAF is referenced by its parent
// as a leaf and is itself dead (never referenced, never emitted).  last[r]: position r ends a leaf.
__global__ __launch_bounds__(BB) void k_radix_tree(const u64* keys, int n, uint32_t* child, uint32_t* parent_inner, uint32_t* covers, uint32_t* last) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) < 0 ? -1 : 1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    covers[i] = (uint32_t)(hi - lo + 1);
    // the left child covers [lo, gamma], the right one [gamma + 1, hi]; an inner child's index is its split-side end
    const bool dead = i != 0 && hi - lo + 1 <= RT_LBVH_LEAF;     // this node lies inside a leaf: its parent does not list it
    if (gamma - lo + 1 <= RT_LBVH_LEAF) {
        child[2 * i] = RT_LEAF_BIT | (uint32_t)lo;
        if (!dead) for (int r = lo; r <= gamma; r++) last[r] = r == gamma ? 1u : 0u;
    } else { child[2 * i] = (uint32_t)gamma; parent_inner[gamma] = (uint32_t)i; }
    if (hi - gamma <= RT_LBVH_LEAF) {
        child[2 * i + 1] = RT_LEAF_BIT | (uint32_t)(gamma + 1);
        if (!dead) for (int r = gamma + 1; r <= hi; r++) last[r] = r == hi ? 1u : 0u;
    } else { child[2 * i + 1] = (uint32_t)(gamma + 1); parent_inner[gamma + 1] = (uint32_t)i; }
    if (i == 0) parent_inner[0] = RT_NONE;
}

// live inner nodes at even depth head a four-wide record
__global__ __launch_bounds__(BB) void k_even_depth(const uint32_t* parent_inner, const uint32_t* covers, uint32_t n_inner, uint32_t* is4) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n_inner) return;
    if (i != 0 && covers[i] <= RT_LBVH_LEAF) { is4[i] = 0u; return; }     // inside a leaf
    uint32_t d = 0;
    for (uint32_t p = i; p != 0; p = parent_inner[p]) d++;
    is4[i] = (d & 1u) ? 0u : 1u;
}

__device__ inline uint32_t wide_children(const uint32_t* child, uint32_t i, uint32_t kids[4]) {
    uint32_t nk = 0;
    for (int side = 0; side < 2; side++) {
        const uint32_t c = child[2 * i + side];
        if (c & RT_LEAF_BIT) kids[nk++] = c;
        else { kids[nk++] = child[2 * c]; kids[nk++] = child[2 * c + 1]; }
    }
    return nk;
}

// The four-wide records' topology: references, parent links, child counts.  Boxes: all empty (the refit fills them).
// A record's pad[0], pad[1] (parent record, slot in it) are written by the PARENT's thread, everything else by its own.
__global__ __launch_bounds__(BB) void k_emit_wide(const uint32_t* child, const uint32_t* parent_inner, const uint32_t* is4, const uint32_t* idx4,
                                                  uint32_t n_inner, DNode* nodes, uint32_t* stack_need) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n_inner || !is4[i]) return;
    const uint32_t me = idx4[i];
    uint32_t kids[4];
    const uint32_t nk = wide_children(child, i, kids);
    DNode* dn = nodes + me;
    uint32_t inner = 0;
    for (uint32_t k = 0; k < 4; k++) {
        dn->minx[k] = dn->miny[k] = dn->minz[k] = INFINITY;
        dn->maxx[k] = dn->maxy[k] = dn->maxz[k] = -INFINITY;
        if (k >= nk) { dn->ref[k] = RT_NO_CHILD; continue; }
        const uint32_t r = kids[k];
        if (r & RT_LEAF_BIT) dn->ref[k] = r;
        else {
            const uint32_t c4 = idx4[r];
            dn->ref[k] = c4;
            nodes[c4].pad[0] = me; nodes[c4].pad[1] = k;
            inner++;
        }
    }
    dn->pad[2] = inner; dn->pad[3] = nk;
    if (i == 0) { dn->pad[0] = RT_NONE; dn->pad[1] = 0; }
    // what a walk can have stacked when it enters this record's children: every record on the way down leaves up to
    // (children - 1) entries behind
    uint32_t held = 0;
    for (uint32_t p = i; p != 0;) {
        const uint32_t up = parent_inner[parent_inner[p]];   // two levels up: the record that adopted this one
        uint32_t tmp[4];
        held += wide_children(child, up, tmp) - 1;
        p = up;
    }
    atomicMax(stack_need, held + nk);
}

__global__ __launch_bounds__(BB) void k_emit_tris(const double* verts, const uint32_t* tri_pos, const uint32_t* order, const uint32_t* last, uint32_t n, DTri* tris, int fresh) {
    const uint32_t r = blockIdx.x * BB + threadIdx.x;
    if (r >= n) return;
    const uint32_t t = fresh ? order[r] : tris[r].orig;
    const double* v0 = verts + 3 * (size_t)tri_pos[3 * (size_t)t];
    const double* v1 = verts + 3 * (size_t)tri_pos[3 * (size_t)t + 1];
    const double* v2 = verts + 3 * (size_t)tri_pos[3 * (size_t)t + 2];
    DTri& d = tris[r];
    for (int k = 0; k < 3; k++) { d.v0[k] = v0[k]; d.e1[k] = v1[k] - v0[k]; d.e2[k] = v2[k] - v0[k]; }   // mesh.rs:26-27
    if (fresh) { d.orig = t; d.last = last ? last[r] : 1u; }
}

// f32 box of a node from its f64 bounds, rounded outward from the slab bounds the reference's test uses:
// centroid -+ size with centroid = (min + max) * 0.5 and size = max - centroid (acceleration.rs:76,135)
__device__ inline void slot_box(DNode* nd, uint32_t k, const double mn[3], const double mx[3]) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        const double cen = (mn[a] + mx[a]) * 0.5;
        const double half = mx[a] - cen;
        lo[a] = __double2float_rd(fmin(mn[a], cen - half));
        hi[a] = __double2float_ru(fmax(mx[a], cen + half));
    }
    nd->minx[k] = lo[0]; nd->miny[k] = lo[1]; nd->minz[k] = lo[2];
    nd->maxx[k] = hi[0]; nd->maxy[k] = hi[1]; nd->maxz[k] = hi[2];
}

__device__ inline void leaf_box(const double* verts, const uint32_t* tri_pos, const DTri* tris, uint32_t rank, double mn[3], double mx[3]) {
    for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; }
    while (true) {
        double tn[3], tx[3];
        tri_box(verts, tri_pos, tris[rank].orig, tn, tx);
        for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], tn[a]); mx[a] = fmax(mx[a], tx[a]); }
        if (tris[rank].last) break;
        ++rank;
    }
}

// Boxes, bottom-up.  arrive[]: zeroed; a record is complete when its inner children and its own thread have arrived.
__global__ __launch_bounds__(BB) void k_refit(const double* verts, const uint32_t* tri_pos, DNode* nodes, uint32_t n_inner, const DTri* tris,
                                              uint32_t* arrive, double* own, double* root_box) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n_inner) return;
    uint32_t cur = i;
    while (true) {
        DNode* nd = nodes + cur;
        const uint32_t needed = nd->pad[2] + 1u;
        const uint32_t before = atomicAdd(&arrive[cur], 1u);
        if (before + 1u < needed) return;
        __threadfence();   // the inner children's boxes (and their entries in this record's slots) are visible from here on
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t r = nd->ref[k];
            if (r == RT_NO_CHILD) continue;
            double cn[3], cx[3];
            if (r & RT_LEAF_BIT) {
                leaf_box(verts, tri_pos, tris, r & ~RT_LEAF_BIT, cn, cx);
                slot_box(nd, k, cn, cx);
            } else {
                const volatile double* b = own + 6 * (size_t)r;
                for (int a = 0; a < 3; a++) { cn[a] = b[a]; cx[a] = b[3 + a]; }
            }
            for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], cn[a]); mx[a] = fmax(mx[a], cx[a]); }
        }
        for (int a = 0; a < 3; a++) { own[6 * (size_t)cur + a] = mn[a]; own[6 * (size_t)cur + 3 + a] = mx[a]; }
        const uint32_t parent = nd->pad[0];
        if (parent == RT_NONE) {
            for (int a = 0; a < 3; a++) { root_box[a] = mn[a]; root_box[3 + a] = mx[a]; }
            return;
        }
        slot_box(nodes + parent, nd->pad[1], mn, mx);
        __threadfence();
        cur = parent;
    }
}

// a mesh whose root is a leaf: its box
__global__ void k_leaf_root(const double* verts, const uint32_t* tri_pos, const DTri* tris, uint32_t root_ref, double* root_box) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double mn[3], mx[3];
    leaf_box(verts, tri_pos, tris, root_ref & ~RT_LEAF_BIT, mn, mx);
    for (int a = 0; a < 3; a++) { root_box[a] = mn[a]; root_box[3 + a] = mx[a]; }
}

#define TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

hipError_t bvh_refit_device(hipStream_t s, const double* d_verts, const uint32_t* d_tri_pos, uint32_t n_records, DNode* nodes, uint32_t n_inner,
                            uint32_t root_ref, DTri* tris, double root_box_out[6]) {
    hipError_t err = hipSuccess;
    char* tmp = nullptr;
    const size_t b_arrive = align256((size_t)n_inner * 4), b_own = align256((size_t)n_inner * 48), b_root = 256;
    TRY(hipMalloc((void**)&tmp, b_arrive + b_own + b_root));
    {
        uint32_t* arrive = reinterpret_cast<uint32_t*>(tmp);
        double* own = reinterpret_cast<double*>(tmp + b_arrive);
        double* root = reinterpret_cast<double*>(tmp + b_arrive + b_own);
        hipLaunchKernelGGL(k_emit_tris, GRID(n_records), dim3(BB), 0, s, d_verts, d_tri_pos, (const uint32_t*)nullptr, (const uint32_t*)nullptr, n_records, tris, 0);
        if (n_inner) {
            TRY(hipMemsetAsync(arrive, 0, b_arrive, s));
            hipLaunchKernelGGL(k_refit, GRID(n_inner), dim3(BB), 0, s, d_verts, d_tri_pos, nodes, n_inner, tris, arrive, own, root);
        } else hipLaunchKernelGGL(k_leaf_root, dim3(1), dim3(64), 0, s, d_verts, d_tri_pos, tris, root_ref, root);
        TRY(hipGetLastError());
        TRY(hipMemcpyAsync(root_box_out, root, 48, hipMemcpyDeviceToHost, s));
        TRY(hipStreamSynchronize(s));
    }
done:
    if (tmp) (void)hipFree(tmp);
    return err;
}

hipError_t bvh_build_device(hipStream_t s, const double* d_verts, const uint32_t* d_tri_pos, uint32_t n_tris, BvhBuildResult* out) {
    hipError_t err = hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    *out = BvhBuildResult();
    const uint32_t n = n_tris, ni = n > 0 ? n - 1 : 0;
    char* tmp = nullptr;
    DNode* nodes = nullptr;
    DTri* tris = nullptr;
    if (n == 0) return hipErrorInvalidValue;
    {
        // scratch, carved from one allocation
        size_t sort_bytes = 0, scan_bytes = 0;
        TRY((hipcub::DeviceRadixSort::SortPairs<u64, uint32_t>(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, (int)n, 0, 63, s)));
        TRY((hipcub::DeviceScan::ExclusiveSum<const uint32_t*, uint32_t*>(nullptr, scan_bytes, nullptr, nullptr, (int)(ni ? ni : 1), s)));
        const size_t b_keys = align256((size_t)n * 8), b_idx = align256((size_t)n * 4), b_ni = align256((size_t)(ni + 1) * 4);
        const size_t b_cub = align256(std::max(sort_bytes, scan_bytes));
        const size_t total = 2 * b_keys + 2 * b_idx + /* child */ 2 * b_ni + /* parent_inner, is4, idx4, covers */ 4 * b_ni + /* last */ b_idx + b_cub + 512;
        TRY(hipMalloc((void**)&tmp, total));
        char* p = tmp;
        auto take = [&](size_t b) { char* r = p; p += b; return r; };
        u64* keys_in = (u64*)take(b_keys); u64* keys = (u64*)take(b_keys);
        uint32_t* idx_in = (uint32_t*)take(b_idx); uint32_t* order = (uint32_t*)take(b_idx);
        uint32_t* child = (uint32_t*)take(2 * b_ni);
        uint32_t* parent_inner = (uint32_t*)take(b_ni); uint32_t* is4 = (uint32_t*)take(b_ni); uint32_t* idx4 = (uint32_t*)take(b_ni);
        uint32_t* last = (uint32_t*)take(b_idx);             // per sorted position: ends a leaf
        uint32_t* covers = (uint32_t*)take(b_ni);
        void* cub = take(b_cub);
        unsigned* small = (unsigned*)take(512);      // [0..5] centroid bounds, [8] stack need
        const unsigned init[9] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};
        TRY(hipMemcpyAsync(small, init, sizeof(init), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_centroid_bounds, GRID(n), dim3(BB), 0, s, d_verts, d_tri_pos, n, small);
        hipLaunchKernelGGL(k_morton, GRID(n), dim3(BB), 0, s, d_verts, d_tri_pos, n, small, keys_in, idx_in);
        TRY((hipcub::DeviceRadixSort::SortPairs<u64, uint32_t>(cub, sort_bytes, keys_in, keys, idx_in, order, (int)n, 0, 63, s)));
        TRY(hipMalloc((void**)&tris, (size_t)n * sizeof(DTri)));
        uint32_t n4 = 0, stack_need = 0;
        if (ni > 0) {
            hipLaunchKernelGGL(k_radix_tree, GRID(ni), dim3(BB), 0, s, keys, (int)n, child, parent_inner, covers, last);
            hipLaunchKernelGGL(k_even_depth, GRID(ni), dim3(BB), 0, s, parent_inner, covers, ni, is4);
            TRY((hipcub::DeviceScan::ExclusiveSum<const uint32_t*, uint32_t*>(cub, scan_bytes, is4, idx4, (int)ni, s)));
            uint32_t last[2] = {0, 0};
            TRY(hipMemcpyAsync(&last[0], idx4 + (ni - 1), 4, hipMemcpyDeviceToHost, s));
            TRY(hipMemcpyAsync(&last[1], is4 + (ni - 1), 4, hipMemcpyDeviceToHost, s));
            TRY(hipStreamSynchronize(s));
            n4 = last[0] + last[1];
            TRY(hipMalloc((void**)&nodes, (size_t)n4 * sizeof(DNode)));
            hipLaunchKernelGGL(k_emit_wide, GRID(ni), dim3(BB), 0, s, child, parent_inner, is4, idx4, ni, nodes, small + 8);
            TRY(hipGetLastError());
            TRY(hipMemcpyAsync(&stack_need, small + 8, 4, hipMemcpyDeviceToHost, s));
        }
        // (a mesh of one triangle has no inner node and no leaf list: its one record ends its leaf)
        hipLaunchKernelGGL(k_emit_tris, GRID(n), dim3(BB), 0, s, d_verts, d_tri_pos, order, ni > 0 ? last : (const uint32_t*)nullptr, n, tris, 1);
        out->root_ref = ni > 0 ? 0u : (RT_LEAF_BIT | 0u);
        out->ms_sort = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        TRY(bvh_refit_device(s, d_verts, d_tri_pos, n, nodes, n4, out->root_ref, tris, out->root_box));
        out->nodes = nodes; out->tris = tris; out->n_inner = n4; out->stack_need = stack_need;
        nodes = nullptr; tris = nullptr;
    }
done:
    if (tmp) (void)hipFree(tmp);
    if (nodes) (void)hipFree(nodes);
    if (tris) (void)hipFree(tris);
    out->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return err;
}
