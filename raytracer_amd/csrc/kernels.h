// Wavefront kernels of the render path.
//
// The reference runs trace_ray recursively per pixel (scene.rs:85-206) and
// quantises every call's result to 8 bits before the caller mixes it in.  Here
// each recursion depth is one "level": a flat array of rays.  Per level,
//   k_trace   finds every ray's closest hit and lists the rays that hit something
//                                                      (scene.rs:97, cast_ray_once)
//   k_shadow  one thread per (hit ray, light): the light's shadow cast
//                                                      (sun.rs:87-94, point.rs:89-99)
//   k_shade   rebuilds the Hit, samples the texture, sums the lights using the
//             shadow results, and appends the child rays to the next level
//                                                      (scene.rs:102-196)
// and, once the deepest level is done, levels are folded back bottom-up by
//   k_resolve which applies the lerps of scene.rs:176-202 to the children's
//             already-quantised colours and truncates to 8 bits itself.
#pragma once
#include "dev_rt.h"
#include "dev_packet.h"

#define RT_BLOCK RT_BLOCK_DIM
#define RT_INACTIVE 0xFFFFFFFEu   // padding slot of an edge tile

// The image is cut into supertiles of S x S tiles (S = 1 << super_shift) of 16 x 16 pixels, numbered row-major; a
// shard owns supertiles shard, shard + n_shards, ...  A supertile is one compact run of S*S*256 slots, tile by tile
// (row-major inside the supertile); tiles of an edge supertile that lie outside the image are all padding.
// Neighbouring tiles share most of the BVH nodes their rays touch: handing a GPU whole 64 x 64 pixel blocks instead
// of every n-th tile keeps those nodes in its L2.
// Row r of supertiles is rotated left by r * row_rot columns before the numbering, chosen so that the column phase a
// shard owns moves by a step coprime to n_shards from one row to the next: over n_shards rows every shard gets every
// phase once (a Latin square) instead of the two phases that 60 supertiles per row and 8 shards would give, and what
// is expensive in the picture (an object, the horizon) is spread over the shards.
struct ShardMap {
    uint32_t shard, n_shards, supers_x, n_supers, super_shift, row_rot;
    uint32_t row_major_out;                       // 1: write the frame row-major; 0: compact slot order
    // a launch may cover this shard of several frames: level 0 holds them back to back, per_frame slots each; in the
    // compact output frame f starts at slot f * frame_stride (>= per_frame: the caller pads every shard to the largest)
    uint32_t per_frame, frame_stride;
    // 1: this shard is one of n_shards PASSES over a frame whose compact output is the whole frame's (shard 0 of 1,
    // whose rows are not rotated): a supertile's slots go where the whole frame keeps that supertile
    uint32_t whole_frame_slots;
};
DEV size_t out_slot(const ShardMap& m, uint32_t i) {
    const uint32_t f = i / m.per_frame, j = i % m.per_frame;
    if (!m.whole_frame_slots) return (size_t)f * m.frame_stride + j;
    const uint32_t bits = 8u + 2u * m.super_shift;   // slots of a supertile
    const uint64_t sid = (uint64_t)(j >> bits) * m.n_shards + m.shard;
    const uint32_t sy = (uint32_t)(sid / m.supers_x);
    const uint32_t sx = (uint32_t)((sid % m.supers_x + (uint64_t)sy * m.row_rot) % m.supers_x);
    return (size_t)f * m.frame_stride + (((size_t)sy * m.supers_x + sx) << bits | (j & ((1u << bits) - 1u)));
}

DEV bool slot_pixel(const ShardMap& m, int32_t vw, int32_t vh, uint32_t slot, int32_t& x, int32_t& y) {
    const uint32_t sh = m.super_shift;
    const uint32_t k = slot >> (8u + 2u * sh), t = (slot >> 8) & ((1u << (2u * sh)) - 1u), j = slot & 255u;
    const uint64_t sid = (uint64_t)k * m.n_shards + m.shard;
    if (sid >= m.n_supers) return false;
    const uint32_t sy = (uint32_t)(sid / m.supers_x);
    const uint32_t sx = (uint32_t)((sid % m.supers_x + (uint64_t)sy * m.row_rot) % m.supers_x);
    const uint32_t tx = (sx << sh) + (t & ((1u << sh) - 1u));
    const uint32_t ty = (sy << sh) + (t >> sh);
    // the 64 slots of one wave cover an 8x8 pixel block (a compact ray packet); four blocks make the tile
    const uint32_t sub = j >> 6, w = j & 63u;
    x = (int32_t)(tx * 16u + (sub & 1u) * 8u + (w & 7u));
    y = (int32_t)(ty * 16u + (sub >> 1) * 8u + (w >> 3));
    return x < vw && y < vh;
}

// FULL = false: origin and direction only -- for kernels of scenes in which nothing reads 1/direction (three f64
// divisions, ray.rs:17-23; its readers are the cube test of aabb.rs:18-67 and the lane-by-lane box walks)
template <bool FULL = true>
DEV Ray load_ray(const DLevel& L, uint32_t i) {
    const D3 o = d3(L.ro[0][i], L.ro[1][i], L.ro[2][i]), d = d3(L.rd[0][i], L.rd[1][i], L.rd[2][i]);
    if (FULL) return ray_new(o, d);
    Ray r = {};
    r.o = o; r.d = d;
    return r;
}
DEV void store_ray(const DLevel& L, uint32_t i, D3 o, D3 d) {
    L.ro[0][i] = o.x; L.ro[1][i] = o.y; L.ro[2][i] = o.z;
    L.rd[0][i] = d.x; L.rd[1][i] = d.y; L.rd[2][i] = d.z;
}

// Level 0 only: primary rays (scene.rs:210-217, camera.rs:55-59).  A launch may cover the same shard of several frames
// that differ in their camera only (a camera path through one scene): frame f = slot / per_frame looks through cams[f];
// cams == nullptr: one frame, the scene's own camera.
__global__ __launch_bounds__(RT_BLOCK) void k_primary(const DScene* Sp, const DCamera* cams, DLevel L, uint32_t n, ShardMap map) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const DScene& S = *Sp;
    const DCamera& cam = cams ? cams[i / map.per_frame] : S.cam;
    int32_t x, y;
    if (!slot_pixel(map, S.vw, S.vh, i % map.per_frame, x, y)) {
        L.hit_obj[i] = RT_INACTIVE;
        store_ray(L, i, d3(0., 0., 0.), d3(0., 0., 1.));
        return;
    }
    L.hit_obj[i] = RT_NONE;
    store_ray(L, i, ld3(cam.cam_o), camera_direction(cam, (double)x, (double)y));
    if (S.has_area) L.key[i] = rng_key_primary(S.area_seed, (unsigned long long)y * (unsigned long long)S.vw + (unsigned long long)x);
}

// A level's size as its kernels see it: `n` is what the launch was sized for -- the size itself when the host knows
// it, otherwise the room the level was given -- and n_dev, when not null, points at the real count in device memory
// (the cursor the level above advanced while spawning).
DEV uint32_t level_size(uint32_t n, const unsigned long long* n_dev) {
    if (!n_dev) return n;
    const unsigned long long real = *n_dev;
    return real < n ? (uint32_t)real : n;
}

DEV uint32_t lane_id() { return __lane_id(); }
DEV unsigned long long lanes_below() { return (1ull << lane_id()) - 1ull; }

#define RT_WAVES (RT_BLOCK / 64)

// Workgroup-wide append: every thread of the block calls it; flagged lanes get consecutive slots from *cursor.
// One atomic per workgroup.  (Two flags at once: all `a` slots of the block come first, then all `b` slots.)
DEV void block_append2(bool a, bool b, unsigned long long* cursor, uint32_t& slot_a, uint32_t& slot_b) {
    __shared__ uint32_t s_cnt[2][RT_WAVES];
    __shared__ uint32_t s_base;
    const unsigned long long ma = __ballot(a), mb = __ballot(b);
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 0) { s_cnt[0][w] = (uint32_t)__popcll(ma); s_cnt[1][w] = (uint32_t)__popcll(mb); }
    __syncthreads();
    uint32_t before_a = 0, before_b = 0, tot_a = 0, tot_b = 0;
#pragma unroll
    for (uint32_t k = 0; k < RT_WAVES; k++) {
        const uint32_t ca = s_cnt[0][k], cb = s_cnt[1][k];
        if (k < w) { before_a += ca; before_b += cb; }
        tot_a += ca; tot_b += cb;
    }
    if (threadIdx.x == 0) s_base = (tot_a + tot_b) ? (uint32_t)atomicAdd(cursor, (unsigned long long)(tot_a + tot_b)) : 0u;
    __syncthreads();
    const uint32_t base = s_base;
    slot_a = base + before_a + (uint32_t)__popcll(ma & lanes_below());
    slot_b = base + tot_a + before_b + (uint32_t)__popcll(mb & lanes_below());
    __syncthreads();   // s_cnt / s_base may be reused by the next call
}
DEV uint32_t block_append(bool a, unsigned long long* cursor) {
    uint32_t sa, sb;
    block_append2(a, false, cursor, sa, sb);
    return sa;
}

// Workgroup-wide sum of up to three small per-thread counts, added to this workgroup's statistics shard.
DEV void block_count3(uint32_t v0, uint32_t v1, uint32_t v2, unsigned long long* c0, unsigned long long* c1, unsigned long long* c2) {
    __shared__ uint32_t s_red[3][RT_WAVES];
    for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); }
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 0) { s_red[0][w] = v0; s_red[1][w] = v1; s_red[2][w] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t0 = 0, t1 = 0, t2 = 0;
        for (uint32_t k = 0; k < RT_WAVES; k++) { t0 += s_red[0][k]; t1 += s_red[1][k]; t2 += s_red[2][k]; }
        if (t0 && c0) atomicAdd(c0, (unsigned long long)t0);
        if (t1 && c1) atomicAdd(c1, (unsigned long long)t1);
        if (t2 && c2) atomicAdd(c2, (unsigned long long)t2);
    }
    __syncthreads();
}
DEV DShard* my_shard(DCounters* C) { return &C->shard[(blockIdx.x + blockIdx.y * 7u) % RT_SHARDS]; }

// Long rays.  A ray that skims a large mesh visits thousands of boxes one after the other, and a launch is not done
// before its slowest ray is: with little other work on the chip (a deep level, a small shard) such rays set the
// frame time.  A walk therefore gives up after a budget of inner nodes and puts its ray on the level's long-ray list;
// a second kernel casts each listed ray again as RT_SEGS pieces in RT_SEGS neighbouring lanes (mesh_traverse_seg)
// and merges their answers.  A level's lists hold every ray of the level (one entry per ray, or per ray and light
// for shadow rays), so a ray that gives up always finds an entry.
#define RT_SEGS 16             /* pieces per long ray = lanes per long ray */
#define RT_LONG_BLOCKS 512     /* grid of the long-ray kernels (they loop over the list) */
struct LongParams { uint32_t budget_lane, budget_wave; };

// what k_trace does with the closest hit of ray i: the full Hit record (normal, uv and far built ONCE here, for k_spawn
// and k_light to read back in coalesced loads instead of each fetching the triangle's indices, normals and texture
// coordinates again), and the entry of the shadow work list (hit point and the offset origin vnear + normal * EPSILON
// of sun.rs:88, point.rs:90) when the hit will be lit
template <bool MESHES = true, bool BOXES = true>
DEV void trace_finish(const DScene& S, DLevel& L, uint32_t i, bool active, const Ray& ray, bool hit, const Closest& c,
                      uint32_t* hitlist, double* hs, uint32_t stride, DLevelCounts* LC) {
    bool lit = false;
    D3 vnear = d3(0., 0., 0.), so = d3(0., 0., 0.);
    if (active) L.hit_obj[i] = c.obj;
    if (active && hit) {
        Hit h;
        closest_hit_fill<MESHES, BOXES>(S, ray, c.obj, c.rank, c.t, c.t2, c.u, c.v, h);
        L.hit_prim[i] = h.prim;
        L.hit_t[i] = h.near; L.hit_t2[i] = h.far; L.hit_u[i] = h.u; L.hit_v[i] = h.v;
        L.nrm[0][i] = h.normal.x; L.nrm[1][i] = h.normal.y; L.nrm[2][i] = h.normal.z;
        if (S.n_lights != 0) {
            const RT_GLOBAL DObject* op = gptr(S.objects) + c.obj;
            if (!(gptr(S.materials)[op->material].emissivity == 1.)) {      // scene.rs:105-107: fully emissive = no lighting
                vnear = h.vnear;
                so = h.vnear + h.normal * RT_EPSILON;
                lit = true;
            }
        }
    }
    const uint32_t slot = block_append(lit, &LC->hit_count);
    if (lit) {
        hitlist[slot] = i;
        hs[slot] = vnear.x; hs[(size_t)stride + slot] = vnear.y; hs[2 * (size_t)stride + slot] = vnear.z;
        hs[3 * (size_t)stride + slot] = so.x; hs[4 * (size_t)stride + slot] = so.y; hs[5 * (size_t)stride + slot] = so.z;
    }
}

// The Hit of ray i as trace_finish stored it (object/mod.rs:18-36)
DEV void load_hit(const DLevel& L, uint32_t i, const Ray& ray, Hit& h) {
    h.normal = d3(L.nrm[0][i], L.nrm[1][i], L.nrm[2][i]);
    h.near = L.hit_t[i]; h.far = L.hit_t2[i];
    h.vnear = along(ray, h.near); h.vfar = along(ray, h.far);
    h.u = L.hit_u[i]; h.v = L.hit_v[i];
    h.prim = L.hit_prim[i];
}

// Closest hit of every ray of the level.  Rays whose hit will be lit are appended (in wave order) to the work list
// of k_shadow, stored plane by plane: hs[k * stride + slot], k = 0..5.
// BUDGET: the lane-by-lane walks give up after lp.budget_lane inner nodes (a separate variant: the bookkeeping costs
// registers and time, and few scenes have rays that need it).
// FAR: see cast_closest_packet.
// (the lane-by-lane variants are asked to fit 4 waves per SIMD: the allocator then finds 118 registers enough, without
// spills, where it takes 156 when left alone; at 5 waves they spill and are 15 % slower.  The packet variants hold most
// of their state in scalar registers and LDS and gain 10 % from a fifth wave.)
#ifndef RT_TRACE_ATTR
#define RT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
#ifndef RT_TRACE_PACKET_ATTR
#define RT_TRACE_PACKET_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
// MESHES = false: a scene of analytic objects only (lane by lane, no budget): the mesh walks stay out of the kernel.
// RECIP = false (deferred packets only): the scene has neither a cube outside the object tree nor an object tree, and
// the rays come without 1/direction: three divisions less per ray, and six registers less to carry through the walk.
template <bool COUNT, bool PACKET, bool BUDGET, bool FAR, bool MESHES, bool DEFER = false, bool RECIP = true>
DEV void trace_body(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, uint32_t* hitlist, double* hs, uint32_t stride,
                    DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    n = level_size(n, n_dev);
    if (blockIdx.x * RT_BLOCK >= n) return;   // (a whole workgroup past the end: launches are sized for the level's room)
    const DScene& S = *Sp;
    Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
    bool active = i < n && L.hit_obj[i] != RT_INACTIVE;
    const uint32_t ii = i < n ? i : 0;
    static_assert(RECIP || (DEFER && !FAR), "only the deferred packets can do without 1/direction");
    Ray ray = load_ray<RECIP>(L, ii);
    Closest c; c.obj = RT_NONE; c.rank = RT_NONE; c.t = 0.; c.t2 = 0.; c.u = 0.f; c.v = 0.f;
    bool hit = false;
    if (PACKET) {
        // the wave's rays are coherent (primary rays of an 8x8 pixel block) and walk the BVH together
        hit = cast_closest_packet<FAR, DEFER, RECIP>(Sp, ray, active, c, vc);
    } else if (active && !BUDGET) {
        hit = MESHES ? cast_closest<FAR>(S, ray, c, vc) : cast_closest_analytic(S, ray, c);
    } else if (active) {
        bool aborted = false;
        hit = cast_closest_seg(S, ray, c, vc, 0, 1, lp.budget_lane, aborted);
        if (aborted) {
            longlist[atomicAdd(&LC->long_trace, 1ull)] = i;   // k_long_trace finishes this ray
            active = false; hit = false;
        }
    }
    trace_finish<MESHES, RECIP>(S, L, i, active, ray, hit, c, hitlist, hs, stride, LC);
    if (COUNT) {
        DShard* sh = my_shard(C);
        if (PACKET) { if (lane_id() == 0) atomicMax(&sh->max_wave_steps, (unsigned long long)(vc.wnodes + vc.wtris)); }
        else atomicMax(&sh->max_lane_steps, (unsigned long long)(vc.nodes + vc.tris));
        block_count3(vc.nodes, vc.tris, vc.units, &sh->nodes_visited, &sh->tris_tested, &sh->units_fetched);
        block_count3(lane_id() == 0 ? vc.wnodes : 0, lane_id() == 0 ? vc.wtris : 0, 0, &sh->wave_nodes, &sh->wave_tris, nullptr);
    }
}
template <bool COUNT, bool PACKET, bool BUDGET, bool FAR = true, bool MESHES = true>
__global__ __launch_bounds__(RT_BLOCK) RT_TRACE_ATTR void k_trace(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, uint32_t* hitlist, double* hs, uint32_t stride,
                                                    DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp) {
    static_assert(!PACKET, "the packet variants are k_trace_packet");
    trace_body<COUNT, false, BUDGET, FAR, MESHES>(Sp, L, n, n_dev, hitlist, hs, stride, C, LC, longlist, lp);
}
template <bool COUNT, bool FAR>
__global__ __launch_bounds__(RT_BLOCK) RT_TRACE_PACKET_ATTR void k_trace_packet(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, uint32_t* hitlist, double* hs, uint32_t stride,
                                                    DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp) {
    trace_body<COUNT, true, false, FAR, true, false>(Sp, L, n, n_dev, hitlist, hs, stride, C, LC, longlist, lp);
}
// packets whose triangle tests and thinly populated nodes wait in queues (scenes without need_far meshes)
#ifndef RT_TRACE_DEFER_ATTR
#define RT_TRACE_DEFER_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
template <bool COUNT, bool RECIP = true>
__global__ __launch_bounds__(RT_BLOCK) RT_TRACE_DEFER_ATTR void k_trace_deferred(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, uint32_t* hitlist, double* hs, uint32_t stride,
                                                    DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp) {
    trace_body<COUNT, true, false, false, true, true, RECIP>(Sp, L, n, n_dev, hitlist, hs, stride, C, LC, longlist, lp);
}

// The listed long rays of a level, RT_SEGS lanes each: lane k walks piece k of the ray through every mesh, the
// pieces' answers are merged (nearest; ties to the lower object index, then to the lower leaf-order index --
// the order cast_ray_once's stable sort gives, scene.rs:57-82), and lane 0 of the group does what k_trace does.
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_long_trace(const DScene* Sp, DLevel L, const uint32_t* longlist, uint32_t* hitlist, double* hs,
                                                         uint32_t stride, DCounters* C, DLevelCounts* LC) {
    const DScene& S = *Sp;
    const uint32_t n_long = (uint32_t)LC->long_trace;
    const uint32_t per_block = RT_BLOCK / RT_SEGS;
    Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
    for (uint32_t base = blockIdx.x * per_block; base < n_long; base += gridDim.x * per_block) {   // uniform per block
        const uint32_t j = base + threadIdx.x / RT_SEGS, k = threadIdx.x % RT_SEGS;
        const bool have = j < n_long;
        const uint32_t i = have ? longlist[j] : 0;
        Ray ray = load_ray(L, i);
        Closest c; bool ab = false;
        bool hit = cast_closest_seg(S, ray, c, vc, k, RT_SEGS, 0, ab);
        // merge over the group's lanes
        for (int off = 1; off < RT_SEGS; off <<= 1) {
            const bool oh = __shfl_xor((int)hit, off) != 0;
            Closest o;
            o.obj = (uint32_t)__shfl_xor((int)c.obj, off); o.rank = (uint32_t)__shfl_xor((int)c.rank, off);
            o.t = __shfl_xor(c.t, off); o.t2 = __shfl_xor(c.t2, off); o.u = __shfl_xor(c.u, off); o.v = __shfl_xor(c.v, off);
            const bool better = oh && (!hit || o.t < c.t || (o.t == c.t && (o.obj < c.obj || (o.obj == c.obj && o.rank < c.rank))));
            if (better) { c = o; hit = true; }
        }
        trace_finish(S, L, i, have && k == 0, ray, hit, c, hitlist, hs, stride, LC);
    }
    if (COUNT) { DShard* sh = my_shard(C); block_count3(vc.nodes, vc.tris, vc.units, &sh->nodes_visited, &sh->tris_tested, &sh->units_fetched); }
}

// The shadow ray of lit hit `hh` towards light `Lt` (sun.rs:87-94, point.rs:68-76,89-99).  cast = 0: the light casts
// no ray here (sun with shadows off; point light beyond max_distance: LightShading::default()).
// FULL = false: origin and direction only, for a cast that never reads 1/direction (three f64 divisions, ray.rs:17-23).
// P: pointer to the light's record -- global, or constant address space where the light is the same for the whole wave
// AREA: the scene has area lights: a record may be one sample point of one (area.rs:117-153: a point light at a point
// drawn from the hit ray's key, keys[hitlist[hh]]) or stand for an area light without samples (no cast).  A sample
// beyond max_distance casts nothing, like a point light; the reference stops the light's loop there, and k_light takes
// the casts of the samples after it off the count again.
template <bool FULL = true, typename P = const RT_GLOBAL DLight*, bool AREA = true>
DEV Ray shadow_ray(P Lt, const double* hs, uint32_t stride, uint32_t hh, uint32_t& cast, bool& limited, double& dist,
                   const uint32_t* hitlist = nullptr, const unsigned long long* keys = nullptr) {
    const D3 vnear = d3(hs[hh], hs[(size_t)stride + hh], hs[2 * (size_t)stride + hh]);
    const D3 so = d3(hs[3 * (size_t)stride + hh], hs[4 * (size_t)stride + hh], hs[5 * (size_t)stride + hh]);
    D3 lv = d3(Lt->v[0], Lt->v[1], Lt->v[2]);
    limited = false; dist = 0.;
    D3 dir;
    if (Lt->kind == RT_DLIGHT_SUN) {
        cast = Lt->shadows ? 1u : 0u;
        dir = -lv;
    } else if (AREA && (Lt->kind == RT_DLIGHT_AREA_EMPTY || (Lt->kind == RT_DLIGHT_AREA_SAMPLE && (!keys || !hitlist)))) {
        cast = 0u;
        dir = d3(0., 0., 1.);
    } else {
        if (AREA && Lt->kind == RT_DLIGHT_AREA_SAMPLE) lv = area_sample(Lt, keys[hitlist[hh]]);
        D3 lvec = lv - vnear;
        dist = magnitude(lvec);
        cast = dist > Lt->max_distance ? 0u : 1u;
        dir = div_by_norm(lvec, dist);
        limited = true;
    }
    if (FULL) return ray_new(so, dir);
    Ray r = {};
    r.o = so; r.d = dir;
    return r;
}

// One thread per (lit hit, light): the light's shadow cast.  Lanes of a wave
// share the light and walk neighbouring hit points, so their shadow rays are coherent.
// occ[light * stride + ray] = 1 when shadowed.
// (held to 7 waves per SIMD, 72 VGPRs: the walk waits on scalar fetches, and waves hide that; 8 waves would fill a
// CU's LDS with shadow workgroups alone and keep the other kernels of the frames in flight off it -- measured
// 3 % slower on the whole frame)
#ifndef RT_SHADOW_ATTR
#define RT_SHADOW_ATTR __attribute__((amdgpu_waves_per_eu(7, 7)))
#endif
// RECIP = false: no cube stands outside the object tree and there is no object tree, so nothing in the cast reads
// 1/direction (aabb.rs:18-67 is its only reader here) and the ray comes without it.
// AREA: see shadow_ray (kernels of their own, k_shadow_area: the other scenes' kernels stay as they are)
template <bool COUNT, bool BUDGET, bool RECIP, bool AREA>
DEV void shadow_body(const DScene* Sp, const uint32_t* hitlist, const double* hs, uint8_t* occ, uint32_t stride,
                     DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp, const unsigned long long* keys) {
    const uint32_t h = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t li = blockIdx.y;
    const DScene& S = *Sp;
    Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
    uint32_t cast = 0;
    const bool have = h < (uint32_t)LC->hit_count;
    if (__ballot(have)) {   // the grid is sized for the worst case: most waves past the hit count have nothing to do
        const uint32_t hh = have ? h : 0;
        bool limited; double dist;
        // (one light per workgroup row: its record comes through the scalar cache)
        const Ray ray = shadow_ray<RECIP, const RT_CONST DLight*, AREA>((const RT_CONST DLight*)uni_ptr(S.lights) + li, hs, stride, hh, cast, limited, dist, hitlist, keys);
        if (!have) cast = 0;
        // the wave's shadow rays leave neighbouring hit points towards one light: walked as a packet
        unsigned long long unresolved = 0;
        const bool blocked = cast_any_packet<BUDGET, RECIP>(Sp, ray, cast != 0, limited, dist, vc, lp.budget_wave, unresolved);
        if (cast) occ[(size_t)li * stride + hitlist[hh]] = blocked ? 1 : 0;
        if (BUDGET && unresolved) {
            // the wave gave up: its open rays go on the long-ray list (k_long_shadow sets their byte when blocked)
            const uint32_t cnt = (uint32_t)__popcll(unresolved);
            unsigned long long base = 0;
            if (lane_id() == 0) base = atomicAdd(&LC->long_shadow, (unsigned long long)cnt);
            base = ((unsigned long long)uni((uint32_t)(base >> 32)) << 32) | uni((uint32_t)base);
            if (lane_in(unresolved)) {
                const unsigned long long e = base + __popcll(unresolved & ((1ull << lane_id()) - 1ull));
                longlist[2 * e] = hh; longlist[2 * e + 1] = li;
            }
        }
    }
    DShard* sh = my_shard(C);
    if (COUNT) {
        if (lane_id() == 0) atomicMax(&sh->max_wave_steps, (unsigned long long)(vc.wnodes + vc.wtris));
        block_count3(cast, vc.nodes, vc.tris, &sh->rays_shadow, &sh->nodes_visited, &sh->tris_tested);
        block_count3(lane_id() == 0 ? vc.wnodes : 0, lane_id() == 0 ? vc.wtris : 0, vc.units, &sh->wave_nodes, &sh->wave_tris, &sh->units_fetched);
    } else {
        // the rays cast, counted per wave: a ballot and one atomic add (no LDS, no barrier at the kernel's end)
        const unsigned long long m = __ballot(cast != 0);
        if (m && lane_id() == 0) atomicAdd(&sh->rays_shadow, (unsigned long long)__popcll(m));
    }
}

template <bool COUNT, bool BUDGET, bool RECIP = true>
__global__ __launch_bounds__(RT_BLOCK) RT_SHADOW_ATTR void k_shadow(const DScene* Sp, const uint32_t* hitlist, const double* hs, uint8_t* occ, uint32_t stride,
                                                                                                          DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp) {
    shadow_body<COUNT, BUDGET, RECIP, false>(Sp, hitlist, hs, occ, stride, C, LC, longlist, lp, nullptr);
}
// the same for a scene with area lights: `keys` are the level's ray keys
template <bool COUNT, bool BUDGET>
__global__ __launch_bounds__(RT_BLOCK) RT_SHADOW_ATTR void k_shadow_area(const DScene* Sp, const uint32_t* hitlist, const double* hs, uint8_t* occ, uint32_t stride,
                                                                                                               DCounters* C, DLevelCounts* LC, uint32_t* longlist, LongParams lp,
                                                                                                               const unsigned long long* keys) {
    shadow_body<COUNT, BUDGET, true, true>(Sp, hitlist, hs, occ, stride, C, LC, longlist, lp, keys);
}

// k_shadow for a scene without meshes: nothing to walk as a packet, so every lane casts on its own, through the
// object tree's wide records with a stack of its own.
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow_lanes(const DScene* Sp, const uint32_t* hitlist, const double* hs, uint8_t* occ, uint32_t stride,
                                                           DCounters* C, const DLevelCounts* LC, const unsigned long long* keys) {
    const uint32_t h = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t li = blockIdx.y;
    const DScene& S = *Sp;
    uint32_t cast = 0;
    if (h < (uint32_t)LC->hit_count) {
        bool limited; double dist;
        const Ray ray = shadow_ray(gptr(S.lights) + li, hs, stride, h, cast, limited, dist, hitlist, keys);
        if (cast) occ[(size_t)li * stride + hitlist[h]] = cast_any_analytic(S, ray, limited, dist) ? 1 : 0;
    }
    block_count3(cast, 0, 0, &my_shard(C)->rays_shadow, nullptr, nullptr);
}

// The listed long shadow rays, RT_SEGS lanes each: any piece that finds a blocker sets the ray's byte.
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_long_shadow(const DScene* Sp, const uint32_t* hitlist, const double* hs, uint8_t* occ, uint32_t stride,
                                                          DCounters* C, const DLevelCounts* LC, const uint32_t* longlist, const unsigned long long* keys) {
    const DScene& S = *Sp;
    const uint32_t n_long = (uint32_t)LC->long_shadow;
    const uint32_t per_block = RT_BLOCK / RT_SEGS;
    Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
    for (uint32_t base = blockIdx.x * per_block; base < n_long; base += gridDim.x * per_block) {
        const uint32_t j = base + threadIdx.x / RT_SEGS, k = threadIdx.x % RT_SEGS;
        if (j >= n_long) continue;
        const uint32_t hh = longlist[2 * (size_t)j], li = longlist[2 * (size_t)j + 1];
        if (li >= S.n_lights) continue;
        uint32_t cast; bool limited; double dist;
        const Ray ray = shadow_ray(gptr(S.lights) + li, hs, stride, hh, cast, limited, dist, hitlist, keys);
        if (cast && cast_any_seg(S, ray, limited, dist, vc, k, RT_SEGS)) occ[(size_t)li * stride + hitlist[hh]] = 1;
    }
    if (COUNT) { DShard* sh = my_shard(C); block_count3(vc.nodes, vc.tris, vc.units, &sh->nodes_visited, &sh->tris_tested, &sh->units_fetched); }
}

DEV bool point_shading(const DLight& Lt, D3 pos, const Ray& ray, const Hit& hit, bool shadowed, double& diffuse, double& specular, double& intensity);
// One light's contribution (lighting/sun.rs:72-97, point.rs:66-105) given its shadow cast's result
DEV void light_shading(const DLight& Lt, const Ray& ray, const Hit& hit, bool shadowed,
                       double& diffuse, double& specular, double& intensity) {
    const double FMAX = 1.7976931348623157e308;
    if (Lt.kind == 0) {
        D3 lvec = -ld3(Lt.v);
        diffuse = clampd(dot(hit.normal, lvec), 0., FMAX);
        D3 halfway = normalize(lvec - ray.d);
        specular = rt_powi(clampd(dot(hit.normal, halfway), 0., FMAX), Lt.specular_power);
        if (Lt.shadows && shadowed) { diffuse *= Lt.shadow_coefficient; specular *= Lt.shadow_coefficient; }
        intensity = Lt.intensity;
        return;
    }
    point_shading(Lt, ld3(Lt.v), ray, hit, shadowed, diffuse, specular, intensity);
}

// point.rs:66-105 (and the loop body of area.rs:117-159) for a light at `pos`; false: the hit is beyond max_distance
DEV bool point_shading(const DLight& Lt, D3 pos, const Ray& ray, const Hit& hit, bool shadowed,
                       double& diffuse, double& specular, double& intensity) {
    const double FMAX = 1.7976931348623157e308;
    D3 lvec = pos - hit.vnear;
    double dist = magnitude(lvec);
    if (dist > Lt.max_distance) { diffuse = 0.; specular = 0.; intensity = 0.; return false; }   // LightShading::default()
    lvec = div_by_norm(lvec, dist);
    diffuse = clampd(dot(hit.normal, lvec), 0., FMAX);
    D3 halfway = normalize(lvec - ray.d);
    specular = rt_powi(clampd(dot(hit.normal, halfway), 0., FMAX), Lt.specular_power);
    if (shadowed) { diffuse *= 0.; specular *= 0.; }
    double q = dist / RT_METER;
    intensity = Lt.intensity / (q * q);
    return true;
}

// What a hit spawns (scene.rs:129-196): the transmitted ray (child 0) and the reflected ray (child 1), the blend
// flags and the reflection weight.  Needs the hit only, not the lighting, so the next level can be traced while this
// level's shadow rays are still in flight.
// MESHREF = false: no mesh of the scene has a refracting material (transparency with ior != 1), so the two helper
// intersects of the refraction branch (scene.rs:143-147) never meet a mesh -- the usual case, and a much lighter
// kernel (no LDS, a third of the registers), which matters because it sits on the frame's critical path beside a
// chip full of shadow waves.
template <bool COUNT, bool MESHREF>
// (the light variant is asked to fit 6 waves per SIMD, 78 registers without spills against 94 when left alone: the
// kernel streams hit records and waits on memory; the variant with the mesh walk is left alone)
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(MESHREF ? 1 : 6, 8))) void k_spawn(const DScene* Sp, DLevel L, DLevel N, uint32_t n, const unsigned long long* n_dev, uint32_t depth,
                                                    uint32_t next_cap, DCounters* C, DLevelCounts* LC, uint32_t* ids_out, ShardMap map) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    n = level_size(n, n_dev);
    if (blockIdx.x * RT_BLOCK >= n) return;
    const size_t io = ids_out ? out_slot(map, i) : 0;   // (ids are written for level 0 only)
    const DScene& S = *Sp;
    bool active = i < n;
    uint32_t obj = active ? L.hit_obj[i] : RT_INACTIVE;
    if (active && obj == RT_INACTIVE) {
        L.result[i] = 0u; L.base_flags[i] = (uint32_t)NF_DONE << 24;
        if (ids_out && depth == 0) { ids_out[2 * io] = RT_NONE; ids_out[2 * io + 1] = RT_NONE; }
        active = false;
    }
    Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
    bool want0 = false, want1 = false, pending = false;
    D3 c0o = d3(0, 0, 0), c0d = d3(0, 0, 1), c1o = d3(0, 0, 0), c1d = d3(0, 0, 1);

    if (active) {
        Ray ray = load_ray(L, i);
        if (obj == RT_NONE) {                                        // scene.rs:99
            L.result[i] = skybox_color(S, ray.d);
            L.base_flags[i] = (uint32_t)NF_DONE << 24;
            if (ids_out && depth == 0) { ids_out[2 * io] = RT_NONE; ids_out[2 * io + 1] = RT_NONE; }
        } else {
            Hit hit;
            load_hit(L, i, ray, hit);
            if (ids_out && depth == 0) { ids_out[2 * io] = obj; ids_out[2 * io + 1] = hit.prim; }
            const DMaterial& mat = S.materials[S.objects[obj].material];
            const uint32_t base_rgb = texture_at(mat, hit.u, hit.v);       // scene.rs:102
            if (mat.emissivity == 1.) {                                    // scene.rs:105-107
                L.result[i] = to_color(to_d3(base_rgb));
                L.base_flags[i] = (uint32_t)NF_DONE << 24;
            } else {
                uint32_t flags = 0;
                double w_r = 0.;
                const double refl = mat.reflectiveness, transp = mat.transparency, ior = mat.ior;
                if (transp > RT_EPSILON && depth < S.max_depth) {          // scene.rs:129-180
                    flags |= NF_TRANSPARENT;
                    if (ior == 1.) {
                        want0 = true; c0o = hit.vfar; c0d = ray.d;         // :134
                    } else {
                        D3 ref_vec;
                        if (refraction_vec(ray.d, hit.normal, 1., ior, ref_vec)) {           // :140 (the reference unwraps)
                            Hit ref_hit;
                            if (object_intersect<MESHREF>(S, obj, ray_new(hit.vnear - ref_vec, ref_vec), ref_hit, vc)) {              // :143
                                Hit exit_hit;
                                if (object_intersect<MESHREF>(S, obj, ray_new(ref_hit.vfar + ref_vec, -ref_vec), exit_hit, vc)) {     // :145-146
                                    D3 exit_ref_vec;
                                    if (refraction_vec(ref_vec, -exit_hit.normal, ior, 1., exit_ref_vec)) {              // :148-153
                                        want0 = true; c0o = ref_hit.vfar + exit_ref_vec * RT_EPSILON; c0d = exit_ref_vec;  // :155
                                    }
                                }
                            }
                        }
                    }
                    if (refl > RT_EPSILON) {                               // :165-177
                        double dd = dot(-ray.d, hit.normal);
                        dd = dd * dd;
                        flags |= NF_T_REFLECT;
                        w_r = 1. - dd;
                        want1 = true;
                    }
                }
                if (refl > RT_EPSILON && depth < S.max_depth && transp < RT_EPSILON) {   // scene.rs:182-196
                    flags |= NF_REFLECT;
                    w_r = refl;
                    want1 = true;
                }
                if (want1) {                                               // ray.reflect, ray.rs:43-46
                    c1o = hit.vnear + hit.normal * RT_EPSILON;
                    c1d = ray.d - hit.normal * (2. * dot(ray.d, hit.normal));
                }
                if (mat.emissivity > 0.) flags |= NF_EMISSIVE;             // scene.rs:200-205
                L.w_t[i] = transp; L.w_r[i] = w_r; L.emis[i] = mat.emissivity;
                L.base_flags[i] = (base_rgb & 0xFFFFFFu) | (flags << 24);   // k_light finishes it
                pending = true;
            }
        }
    }
    // append the children to the next level (whole wave takes part)
    uint32_t s0, s1;
    block_append2(want0, want1, &LC->next_count, s0, s1);
    if (want0 && s0 >= next_cap) want0 = false;
    if (want1 && s1 >= next_cap) want1 = false;
    if (pending) {
        L.child0[i] = want0 ? s0 : RT_NONE;
        L.child1[i] = want1 ? s1 : RT_NONE;
    }
    if (want0) { store_ray(N, s0, c0o, c0d); N.hit_obj[s0] = RT_NONE; }
    if (want1) { store_ray(N, s1, c1o, c1d); N.hit_obj[s1] = RT_NONE; }
    if (S.has_area && (want0 || want1)) {
        const unsigned long long key = L.key[i];
        if (want0) N.key[s0] = rng_key_child(key, RT_RNG_CHILD_TRANSMIT);
        if (want1) N.key[s1] = rng_key_child(key, RT_RNG_CHILD_REFLECT);
    }
    DShard* sh = my_shard(C);
    block_count3(want0 ? 1u : 0u, want1 ? 1u : 0u, 0, &sh->rays_transmit, &sh->rays_reflect, nullptr);
    if (COUNT) block_count3(vc.nodes, vc.tris, vc.units, &sh->nodes_visited, &sh->tris_tested, &sh->units_fetched);
}

// How many rays the level may spawn at most: one per hit on a transparent material, one per hit on a reflective one
// (scene.rs:129,165,182; fewer are spawned when a refraction fails, scene.rs:140-153).  Lets the host give the next
// level exactly the room it can need before k_spawn runs.
__global__ __launch_bounds__(RT_BLOCK) void k_count_children(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, uint32_t depth, DLevelCounts* LC) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    n = level_size(n, n_dev);
    if (blockIdx.x * RT_BLOCK >= n) return;
    const DScene& S = *Sp;
    uint32_t c = 0;
    if (i < n && depth < S.max_depth) {
        const uint32_t obj = L.hit_obj[i];
        if (obj != RT_NONE && obj != RT_INACTIVE) {
            const DMaterial& mat = S.materials[S.objects[obj].material];
            if (!(mat.emissivity == 1.)) {
                const bool t = mat.transparency > RT_EPSILON;
                c = (t ? 1u : 0u) + ((mat.reflectiveness > RT_EPSILON && (t || mat.transparency < RT_EPSILON)) ? 1u : 0u);
            }
        }
    }
    block_count3(c, 0, 0, &LC->child_bound, nullptr, nullptr);
}

// The lit colour of every hit that k_spawn left pending (scene.rs:102-122), from the shadow casts' results.  A hit
// without children is final here; the others keep the colour for k_resolve.
// (64 registers, 8 waves per SIMD, no spills; 80 when left alone)
// AREA: the scene has area lights (a variant of its own).  The records of one area light's samples stand in a row;
// their shadings are summed as area.rs:161-165 sums them (each of the three sums in sample order) and divided by the
// number of samples -- none: 0 / 0, as there.  The first sample beyond max_distance makes the light's shading
// LightShading::default() (area.rs:126-128) and ends its loop there: the samples after it cast nothing in the
// reference, so what k_shadow counted for them is taken off again.
template <bool AREA>
DEV void light_body(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, const uint8_t* occ, uint32_t stride, DCounters* C) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= level_size(n, n_dev)) return;
    const uint32_t bf = L.base_flags[i];
    const uint32_t flags = bf >> 24;
    if (flags & NF_DONE) return;
    const DScene& S = *Sp;
    const uint32_t obj = L.hit_obj[i];
    Ray ray = load_ray(L, i);
    Hit hit;
    load_hit(L, i, ray, hit);
    const uint32_t base_rgb = bf & 0xFFFFFFu;
    D3 color = to_d3(base_rgb);
    D3 sum = ld3(S.ambient);                                   // scene.rs:110-120
    unsigned long long uncast = 0;
    for (uint32_t li = 0; li < S.n_lights; li++) {
        const DLight& Lt = S.lights[li];
        double diff = 0., spec = 0., inten = 0.;
        if (AREA && Lt.kind == RT_DLIGHT_AREA_EMPTY) {
            diff = 0. / 0.; spec = diff; inten = diff;
        } else if (AREA && Lt.kind == RT_DLIGHT_AREA_SAMPLE) {
            // (the row's records differ in area_it only)
            const unsigned long long key = L.key[i];
            double sd = 0., ss = 0., si = 0.;
            bool far = false;
            const uint32_t rows = Lt.area_n < S.n_lights - li ? Lt.area_n : S.n_lights - li;
            for (uint32_t it = 0; it < rows; it++) {
                const DLight& Ls = S.lights[li + it];
                double d1, s1, i1;
                const bool near = point_shading(Ls, area_sample(&Ls, key), ray, hit, occ[(size_t)(li + it) * stride + i] != 0, d1, s1, i1);
                if (far) { if (near) uncast++; }
                else if (!near) far = true;
                else { sd += d1; ss += s1; si += i1; }
            }
            if (!far) { const double cnt = (double)rows; diff = sd / cnt; spec = ss / cnt; inten = si / cnt; }
            li += rows - 1;
        } else
            light_shading(Lt, ray, hit, occ[(size_t)li * stride + i] != 0, diff, spec, inten);
        D3 lcol = ld3(Lt.col);
        D3 dv = lcol * diff;
        D3 sv = lcol * (spec * Lt.specular_strength);
        sum = sum + (dv + sv) * inten;
    }
    color = color * sum;                                       // scene.rs:122
    if (L.child0[i] == RT_NONE && L.child1[i] == RT_NONE) {
        // no child calls: the result is final now
        if (flags & NF_TRANSPARENT) color = vlerp(color, color, L.w_t[i]);     // :179 with transparency_color == color
        if (flags & NF_EMISSIVE) color = vlerp(color, to_d3(base_rgb), L.emis[i]);
        L.result[i] = to_color(color);
        L.base_flags[i] = (uint32_t)NF_DONE << 24;
    } else {
        L.lit[0][i] = color.x; L.lit[1][i] = color.y; L.lit[2][i] = color.z;
    }
    if (AREA && uncast) atomicAdd(&my_shard(C)->rays_shadow, 0ull - uncast);   // (the shards are summed modulo 2^64)
}
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_light(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, const uint8_t* occ, uint32_t stride) {
    light_body<false>(Sp, L, n, n_dev, occ, stride, nullptr);
}
__global__ __launch_bounds__(RT_BLOCK) void k_light_area(const DScene* Sp, DLevel L, uint32_t n, const unsigned long long* n_dev, const uint8_t* occ, uint32_t stride, DCounters* C) {
    light_body<true>(Sp, L, n, n_dev, occ, stride, C);
}

// Folds one level into its parents' view: scene.rs:176-205 on the children's
// 8-bit results.  Level 0 also writes the frame.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(const DScene* Sp, DLevel L, DLevel N, uint32_t n, const unsigned long long* n_dev, int level0,
                                                      uint8_t* rgb_out, ShardMap map) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= level_size(n, n_dev)) return;
    const uint32_t bf = L.base_flags[i];
    const uint32_t flags = bf >> 24;
    uint32_t res;
    if (flags & NF_DONE) res = L.result[i];
    else {
        D3 color = d3(L.lit[0][i], L.lit[1][i], L.lit[2][i]);
        const double w_r = L.w_r[i];
        const uint32_t c0 = L.child0[i], c1 = L.child1[i];
        if (flags & NF_TRANSPARENT) {
            D3 tc = color;
            if (c0 != RT_NONE) tc = to_d3(N.result[c0]);                                       // :136,:158
            if ((flags & NF_T_REFLECT) && c1 != RT_NONE) tc = vlerp(tc, to_d3(N.result[c1]), w_r);   // :176
            color = vlerp(color, tc, L.w_t[i]);                                                // :179
        }
        if ((flags & NF_REFLECT) && c1 != RT_NONE) color = vlerp(color, to_d3(N.result[c1]), w_r);   // :195
        if (flags & NF_EMISSIVE) color = vlerp(color, to_d3(bf & 0xFFFFFFu), L.emis[i]);       // :202
        res = to_color(color);
        L.result[i] = res;
    }
    if (level0) {
        size_t o = out_slot(map, i);
        if (map.row_major_out) {
            const DScene& S = *Sp;
            int32_t x, y;
            if (!slot_pixel(map, S.vw, S.vh, i, x, y)) return;
            o = (size_t)y * (size_t)S.vw + (size_t)x;
        }
        rgb_out[3 * o] = (uint8_t)(res & 255u); rgb_out[3 * o + 1] = (uint8_t)((res >> 8) & 255u); rgb_out[3 * o + 2] = (uint8_t)((res >> 16) & 255u);
    }
}

// ids (2 x u32 per slot) compact -> row-major, for rt_render
__global__ __launch_bounds__(RT_BLOCK) void k_ids_rowmajor(const DScene* Sp, const uint32_t* compact, uint32_t* out, uint32_t n, ShardMap map) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const DScene& S = *Sp;
    int32_t x, y;
    if (!slot_pixel(map, S.vw, S.vh, i, x, y)) return;
    size_t o = (size_t)y * (size_t)S.vw + (size_t)x;
    out[2 * o] = compact[2 * (size_t)i]; out[2 * o + 1] = compact[2 * (size_t)i + 1];
}

// A finished frame out of device memory into host memory the device can write (the library's pinned staging buffer),
// by the device itself: every workgroup moves RT_COPY_BLOCK bytes as 16-byte words and then raises its flag -- a word of
// its own in host memory, stored with release semantics at system scope, so that a host thread that sees the flag
// sees the bytes.  Host threads move a block's bytes on into the caller's buffer as soon as its flag is up: no copy
// commands, no events, nothing of the runtime between the last kernel of the frame and the first byte in host memory.
#define RT_COPY_BLOCK 32768u
__global__ __launch_bounds__(256) void k_copy_out(const uint8_t* src, uint8_t* dst_host, unsigned long long bytes, unsigned int* flags) {
    const unsigned long long base = (unsigned long long)blockIdx.x * RT_COPY_BLOCK;
    const unsigned long long end = bytes < base + RT_COPY_BLOCK ? bytes : base + RT_COPY_BLOCK;
    for (unsigned long long off = base + threadIdx.x * 16ull; off + 16ull <= end; off += 256ull * 16ull)
        *reinterpret_cast<uint4*>(dst_host + off) = *reinterpret_cast<const uint4*>(src + off);
    const unsigned long long tail = end & ~15ull;       // (only the last block can end off a 16-byte boundary)
    if (tail + threadIdx.x < end) dst_host[tail + threadIdx.x] = src[tail + threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Gathered compact shard buffers -> row-major images (rt_detile, rt_detile_frames).  Layout of `gathered`:
// [shard][frame][slot], every (shard, frame) run padded to slots_per_shard slots; image f starts at f * vw * vh elements.
__global__ __launch_bounds__(RT_BLOCK) void k_detile(const uint8_t* gathered, uint8_t* images, int32_t vw, int32_t vh, uint32_t n_shards, uint32_t n_frames,
                                                     unsigned long long slots_per_shard, uint32_t elem_bytes, ShardMap m) {
    const unsigned long long g = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
    const unsigned long long total = slots_per_shard * n_shards * n_frames;
    if (g >= total) return;
    const unsigned long long run = g / slots_per_shard;
    m.shard = (uint32_t)(run / n_frames);
    const uint32_t frame = (uint32_t)(run % n_frames);
    // The 64 slots of a wave are one 8x8 pixel block (while the caller's slots_per_shard is a multiple of 64, as a
    // shard's own slot count is, a wave never straddles two shards or frames): 192 consecutive bytes in, eight rows of 24 bytes out.  Where the block lies inside
    // the frame and everything is dword-aligned, 48 lanes move one dword each instead of 64 lanes three bytes each.
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long g0 = g - lane;
    int32_t x0, y0;
    const bool inside0 = slot_pixel(m, vw, vh, (uint32_t)(g0 % slots_per_shard), x0, y0);
    if (elem_bytes == 3u && (slots_per_shard & 63ull) == 0 && (vw & 3) == 0 && ((((size_t)gathered) | ((size_t)images)) & 3u) == 0 && inside0 &&
        x0 + 8 <= vw && y0 + 8 <= vh) {
        if (lane < 48u) {
            const uint32_t r = lane / 6u, c = lane % 6u;
            const uint32_t v = reinterpret_cast<const uint32_t*>(gathered + g0 * 3u)[lane];
            reinterpret_cast<uint32_t*>(images + (((size_t)frame * (size_t)vh + (size_t)(y0 + (int32_t)r)) * (size_t)vw + (size_t)x0) * 3u)[c] = v;
        }
        return;
    }
    int32_t x, y;
    if (!slot_pixel(m, vw, vh, (uint32_t)(g % slots_per_shard), x, y)) return;
    const uint8_t* src = gathered + g * elem_bytes;
    uint8_t* dst = images + (((size_t)frame * (size_t)vh + (size_t)y) * (size_t)vw + (size_t)x) * elem_bytes;
    for (uint32_t b = 0; b < elem_bytes; b++) dst[b] = src[b];
}

// Probe rays (rt_cast_rays): Scene::cast_ray_once with the full Hit
struct DProbeHit { uint32_t object, prim; double near_t, far_t; double normal[3], vnear[3], vfar[3]; float uv[2]; };
__global__ __launch_bounds__(RT_BLOCK) void k_probe(const DScene* Sp, const double* origins, const double* dirs, DProbeHit* out, uint32_t n) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const DScene& S = *Sp;
    Ray ray = ray_new(ld3(origins + 3 * (size_t)i), ld3(dirs + 3 * (size_t)i));
    DProbeHit o;
    o.object = RT_NONE; o.prim = RT_NONE; o.near_t = 0.; o.far_t = 0.;
    for (int k = 0; k < 3; k++) { o.normal[k] = 0.; o.vnear[k] = 0.; o.vfar[k] = 0.; }
    o.uv[0] = 0.f; o.uv[1] = 0.f;
    // a probe reports Hit::far for every object kind, so meshes are walked in two-nearest mode
    bool any = false; uint32_t best = RT_NONE; Hit bh;
    for (uint32_t oi = 0; oi < S.n_objects; oi++) {
        Hit h;
        Visits vc; vc.nodes = 0; vc.tris = 0; vc.wnodes = 0; vc.wtris = 0; vc.units = 0;
        if (object_intersect(S, oi, ray, h, vc) && (!any || h.near < bh.near)) { any = true; best = oi; bh = h; }
    }
    if (any) {
        o.object = best; o.prim = bh.prim; o.near_t = bh.near; o.far_t = bh.far;
        o.normal[0] = bh.normal.x; o.normal[1] = bh.normal.y; o.normal[2] = bh.normal.z;
        o.vnear[0] = bh.vnear.x; o.vnear[1] = bh.vnear.y; o.vnear[2] = bh.vnear.z;
        o.vfar[0] = bh.vfar.x; o.vfar[1] = bh.vfar.y; o.vfar[2] = bh.vfar.z;
        o.uv[0] = bh.u; o.uv[1] = bh.v;
    }
    out[i] = o;
}
