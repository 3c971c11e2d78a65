// Wave-coherent ray packets: the 64 rays of a wave walk a mesh BVH TOGETHER.
//
// Primary rays of an 8x8 pixel block and the shadow rays of neighbouring hit points towards one light pass
// through nearly the same nodes.  Walked lane by lane, every lane fetches its own copy of every node through
// the vector memory path, whose L1-to-register bandwidth (64 B/clk per CU) then bounds the kernel.  Here the
// wave keeps ONE traversal state -- current node, stack, and per node the mask of lanes whose ray entered it --
// in scalar registers and LDS; node and triangle records are fetched once per wave with scalar loads, and each
// lane only runs the box / triangle arithmetic against its own ray.  A node is visited when ANY lane needs it.
//
// Results are those of the per-lane walk (and of the reference, mesh.rs:326-439): a lane tests exactly the
// triangles of leaves whose box chain its own ray passed, with the same exact f64 triangle test, keeps the
// nearest hit with ties to the lower leaf-order index, and only the ORDER of the tests differs.
#pragma once
#include "dev_rt.h"

#define RT_CONST __attribute__((address_space(4)))   /* constant address space: uniform addresses load through the scalar cache */
typedef unsigned int __attribute__((ext_vector_type(4))) rt_u4;

#define RT_PSTACK 64   /* packet stack entries per wave (16 bytes each); upload refuses a mesh whose walk could need more */
__shared__ rt_u4 rt_pstack[(RT_BLOCK_DIM / 64) * RT_PSTACK];

DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEV unsigned long long uni64(unsigned long long v) { return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
template <typename T> DEV const T* uni_ptr(const T* p) { return (const T*)uni64((unsigned long long)p); }
DEV bool lane_in(unsigned long long mask) { return (mask >> __lane_id()) & 1ull; }
DEV float asf(uint32_t u) { return __uint_as_float(u); }
DEV float first_lane_f(float v, unsigned long long mask) {   // v of the lowest lane set in mask (mask != 0)
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), __ffsll((long long)mask) - 1));
}
DEV double mkd(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

// one node record, wave-uniform (scalar registers)
struct UNode { rt_u4 minx, miny, minz, maxx, maxy, maxz, ref; };
DEV UNode load_unode(const DNode* nodes, uint32_t idx) {
    const RT_CONST rt_u4* p = (const RT_CONST rt_u4*)(nodes + idx);
    UNode n;
    n.minx = p[0]; n.miny = p[1]; n.minz = p[2]; n.maxx = p[3]; n.maxy = p[4]; n.maxz = p[5]; n.ref = p[6];
    return n;
}
DEV TriData load_utri(const DTri* tris, uint32_t rank) {
    const RT_CONST rt_u4* p = (const RT_CONST rt_u4*)(tris + rank);
    const rt_u4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriData t;
    t.v0 = d3(mkd(a.x, a.y), mkd(a.z, a.w), mkd(b.x, b.y));
    t.e1 = d3(mkd(b.z, b.w), mkd(c.x, c.y), mkd(c.z, c.w));
    t.e2 = d3(mkd(d.x, d.y), mkd(d.z, d.w), mkd(e.x, e.y));
    t.orig = e.z; t.last = e.w;
    return t;
}

// lanes (of the whole wave) whose ray meets the box before `limit`: entry <= exit, exit >= 0, entry <= limit.
// Every value is finite (finite boxes, clamped 1/direction), so the ordered compare needs no NaN case.
#define RT_FCMP_OLE 5
// SIGNED: distances of either sign count (entry <= exit, exit >= 0, entry <= limit, nothing clamped at the origin).
template <bool SIGNED = false>
DEV unsigned long long box_lanes(float tn, float tf, float limit) {
    if (SIGNED) return __builtin_amdgcn_fcmpf(tn, fminf(tf, limit), RT_FCMP_OLE) & __builtin_amdgcn_fcmpf(0.f, tf, RT_FCMP_OLE);
    return __builtin_amdgcn_fcmpf(fmaxf(tn, 0.f), fminf(tf, limit), RT_FCMP_OLE);
}

// uniform compare-exchange on (key, ref, mask)
#define RT_UCSWAP(ka, ra, ma, kb, rb, mb)                                     \
    do {                                                                      \
        if ((kb) < (ka)) {                                                    \
            const float k_ = (ka); (ka) = (kb); (kb) = k_;                    \
            const uint32_t r_ = (ra); (ra) = (rb); (rb) = r_;                 \
            const unsigned long long m_ = (ma); (ma) = (mb); (mb) = m_;       \
        }                                                                     \
    } while (0)

// the same with each lane's own value t of the two children following along
#define RT_UCSWAP_T(ka, ra, ma, ta, kb, rb, mb, tb)                           \
    do {                                                                      \
        if ((kb) < (ka)) {                                                    \
            const float k_ = (ka); (ka) = (kb); (kb) = k_;                    \
            const uint32_t r_ = (ra); (ra) = (rb); (rb) = r_;                 \
            const unsigned long long m_ = (ma); (ma) = (mb); (mb) = m_;       \
            const float t_ = (ta); (ta) = (tb); (tb) = t_;                    \
        }                                                                     \
    } while (0)

struct PStack {
    rt_u4* base;   // this wave's entries
    int sp;        // uniform
    DEV void push(uint32_t ref, unsigned long long mask) {
        rt_u4 e; e.x = ref; e.y = (uint32_t)mask; e.z = (uint32_t)(mask >> 32); e.w = 0;
        base[sp] = e;   // every lane writes the same value to the same address
        ++sp;
    }
    DEV void pop(uint32_t& ref, unsigned long long& mask) {
        --sp;
        const rt_u4 e = base[sp];
        ref = uni(e.x);
        mask = ((unsigned long long)uni(e.z) << 32) | uni(e.y);
    }
};

// Closest hit of a whole wave's rays in one mesh: nearest hit with t < bound per lane, ties to the lower leaf-order
// index.  Every lane of the wave must call this; `active` says whether the lane carries a ray.
__device__ bool packet_traverse(const DMesh& Mg, const Ray& ray, bool active, double bound, MeshHit& mh, Visits& vc) {
    mh.n = 0; mh.t0 = bound; mh.t1 = bound; mh.r0 = 0;
    const DNode* nodes = uni_ptr(Mg.nodes);
    const DTri* tris = uni_ptr(Mg.tris);
    float limit = limit_up(bound);
    const PRay pr = pray_new(ray, asf(uni(__float_as_uint(Mg.coord_max))));

    float tn_root;
    const bool root_hit = active && node_box(Mg.root_min, Mg.root_max, ray, tn_root) && !(tn_root > limit);
    unsigned long long curmask = __ballot(root_hit);
    if (!curmask) return false;
    uint32_t cur = uni(Mg.root_ref);
    PStack st; st.base = &rt_pstack[(threadIdx.x >> 6) * RT_PSTACK]; st.sp = 0;

    while (true) {
        if (cur & RT_LEAF_BIT) {
            // ---- a leaf: every triangle, against the lanes that entered the leaf's box
            uint32_t rank = cur & ~RT_LEAF_BIT;
            const bool in = lane_in(curmask);
            while (true) {
                const TriData T = load_utri(tris, rank);
                double t; float u, v;
                if (in) vc.tris++;
                vc.wtris++;
                if (__lane_id() == 0) vc.units += 5;
                if (in && tri_test(T, ray, t, u, v) && (t < mh.t0 || (t == mh.t0 && rank < mh.r0))) {
                    mh.t0 = t; mh.r0 = rank; mh.u = u; mh.v = v; mh.n = 1;
                    limit = limit_up(t);
                }
                if (T.last) break;
                ++rank;
            }
            cur = RT_POP;
        } else {
            // ---- an inner node: four boxes against every lane that entered the node
            const UNode n = load_unode(nodes, cur);
            if (lane_in(curmask)) vc.nodes++;
            vc.wnodes++;
            if (__lane_id() == 0) vc.units += 7;
            float t0, t1, t2, t3, f0, f1, f2, f3;
            RT_CHILD_PAIR(n, 0, t0, f0, t1, f1);
            RT_CHILD_PAIR(n, 2, t2, f2, t3, f3);
            // the loop runs with every lane of the wave enabled, so a compare's lane mask is already the ballot
            const unsigned long long inmask = curmask;
            unsigned long long m0 = box_lanes(t0, f0, limit) & inmask, m1 = box_lanes(t1, f1, limit) & inmask;
            unsigned long long m2 = box_lanes(t2, f2, limit) & inmask, m3 = box_lanes(t3, f3, limit) & inmask;
            uint32_t r0 = n.ref[0], r1 = n.ref[1], r2 = n.ref[2], r3 = n.ref[3];
            if (r2 == RT_NO_CHILD) m2 = 0;
            if (r3 == RT_NO_CHILD) m3 = 0;
            // visit order: by the entry distance of the first lane in each child (rays of a packet agree mostly)
            const float INF = asf(0x7F800000u);
            float k0 = m0 ? first_lane_f(t0, m0) : INF, k1 = m1 ? first_lane_f(t1, m1) : INF;
            float k2 = m2 ? first_lane_f(t2, m2) : INF, k3 = m3 ? first_lane_f(t3, m3) : INF;
            RT_UCSWAP(k0, r0, m0, k1, r1, m1); RT_UCSWAP(k2, r2, m2, k3, r3, m3);
            RT_UCSWAP(k0, r0, m0, k2, r2, m2); RT_UCSWAP(k1, r1, m1, k3, r3, m3);
            RT_UCSWAP(k1, r1, m1, k2, r2, m2);
            // children nobody entered have an empty mask; the first non-empty one is next, the rest are stacked
            cur = RT_POP;
            if (m3) { cur = r3; curmask = m3; }
            if (m2) { if (cur != RT_POP) st.push(cur, curmask); cur = r2; curmask = m2; }
            if (m1) { if (cur != RT_POP) st.push(cur, curmask); cur = r1; curmask = m1; }
            if (m0) { if (cur != RT_POP) st.push(cur, curmask); cur = r0; curmask = m0; }
        }
        if (cur == RT_POP) {
            if (st.sp == 0) break;
            st.pop(cur, curmask);
        }
    }
    return mh.n > 0;
}

// ---- shadow casts with the triangle tests set aside
//
// In the walk above a leaf costs the wave one full triangle test per triangle, whatever the number of lanes whose
// ray entered the leaf -- on a terrain mesh about 8 of 64, so most of the arithmetic of a shadow cast runs on idle
// lanes.  A shadow cast only asks WHETHER something blocks the ray, so the tests can wait: the walk notes (lane,
// leaf) pairs in an LDS queue, and whenever 64 pairs have gathered, every lane of the wave takes one pair, reads
// that ray (parked in LDS at the start) and the leaf's triangles (vector loads), and runs the same exact test.  A
// hit clears the owner's bound word; owners look at their word after each batch and leave the packet.
// What is tested is unchanged (each ray against the triangles of the leaves its box chain reached, until the ray is
// known to be blocked), only later.
#ifndef RT_QFLUSH
#define RT_QFLUSH 48   /* pairs waiting at which the walk stops for a round (a little short of a full round: the rays found blocked leave the packet sooner, 4 % fewer node steps) */
#endif
#define RT_QCAP 304                       /* fewer than RT_QFLUSH pairs left over + the up to 4 x 64 that one node step, or one round of node tests, can add */
static_assert(RT_QCAP >= RT_QFLUSH + 255, "the pair queue must take one step's pairs on top of what a round leaves waiting");
#define RT_PACKET_MAX_PRIMS (1u << 26)    /* a pair is (leaf's first triangle << 6) | lane in one word */
#define RT_PWAVES (RT_BLOCK_DIM / 64)
__shared__ double rt_qray[RT_PWAVES * 6 * 64];                // [wave][o.x o.y o.z d.x d.y d.z][lane]
__shared__ unsigned long long rt_qbound[RT_PWAVES * 64];      // bits of the lane's distance limit, + 1; 0 = blocked / no ray
__shared__ uint32_t rt_qpair[RT_PWAVES * RT_QCAP];
// (lane, inner node) pairs: the thin ends of the walk, see node_pairs()
#ifndef RT_SPARSE
#define RT_SPARSE 16    /* a child entered by at most this many lanes leaves the packet walk */
#endif
#ifndef RT_SPARSE_CLOSEST
#define RT_SPARSE_CLOSEST RT_SPARSE   /* the same for a closest-hit walk */
#endif
#ifndef RT_QFLUSH_OBJ
#define RT_QFLUSH_OBJ RT_QFLUSH   /* the same for the closest-hit walk of an object tree */
#endif
#ifndef RT_NCAP
#define RT_NCAP 236   /* (with the other arrays of a shadow workgroup, the counting variants' included: 23,024 bytes of the 23,040 that 18 LDS granules hold) */
#endif
#ifndef RT_NFLUSH
#define RT_NFLUSH 48    /* node pairs waiting at which the packet walk stops for a round of node tests */
#endif
__shared__ uint32_t rt_qnode[RT_PWAVES * RT_NCAP];
// how much of the node queue a walk uses (<= RT_NCAP; rt_context_create lowers it on request, so that tests can make
// the queue overflow at will)
__device__ uint32_t rt_node_cap = RT_NCAP;
// closest-hit walks only: the entry distance of each queued node pair (a pair whose node lies beyond the ray's best
// hit by the time its turn comes is dropped unread), and per ray the best hit's triangle and barycentrics
__shared__ float rt_qntn[RT_PWAVES * RT_NCAP];
__shared__ uint32_t rt_qrank[RT_PWAVES * 64];
__shared__ float2 rt_quv[RT_PWAVES * 64];

DEV bool in_mask(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }   // this lane's bit, for free
DEV uint32_t lanes_below(unsigned long long mask) {   // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// This lane's number, worked out on the spot: where registers are short the compiler otherwise keeps the (loop
// invariant) value in scratch memory and fetches it back for every use.
DEV uint32_t lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
DEV void wave_lds_sync() {   // LDS accesses of one wave execute in order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A wave-uniform stack in three vector registers: entry i lives in lane i (v_writelane / v_readlane with a uniform
// lane number), so a push or pop is three register moves -- no LDS traffic and no wait on the way to the next
// node fetch.  64 entries; upload refuses a mesh whose walk could need more.
extern "C" __device__ int rt_writelane_i32(int value, int lane, int into) __asm("llvm.amdgcn.writelane.i32");
struct RStack {
    int ref, lo, hi;
    uint32_t sp;
    DEV void push(uint32_t r, unsigned long long m) {
        ref = rt_writelane_i32((int)r, (int)sp, ref);
        lo = rt_writelane_i32((int)(uint32_t)m, (int)sp, lo);
        hi = rt_writelane_i32((int)(uint32_t)(m >> 32), (int)sp, hi);
        ++sp;
    }
    DEV unsigned long long top_mask() const {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(hi, (int)sp - 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, (int)sp - 1);
    }
    DEV uint32_t top_ref() const { return (uint32_t)__builtin_amdgcn_readlane(ref, (int)sp - 1); }
};

// Parks the wave's rays.  bound: hits with t <= bound block the ray (t > 0 always, so the bit patterns order like the
// values; a NaN or non-positive bound can never be met).
DEV void park_rays(const Ray& ray, bool active, double bound) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    double* qr = rt_qray + w * 6 * 64;
    qr[l] = ray.o.x; qr[64 + l] = ray.o.y; qr[128 + l] = ray.o.z;
    qr[192 + l] = ray.d.x; qr[256 + l] = ray.d.y; qr[320 + l] = ray.d.z;
    unsigned long long b = 0;
    if (active) b = bound > 0. ? (unsigned long long)__double_as_longlong(bound) + 1ull : 1ull;
    rt_qbound[w * 64 + l] = b;
    wave_lds_sync();
}
// The same for a closest-hit walk: the bound word holds the bits of the best distance so far (hits must be nearer,
// or as near with a lower triangle position: the outside bound enters with position 0, which nothing beats);
// 0 = no ray.
DEV void park_rays_closest(const Ray& ray, bool active, double bound) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    double* qr = rt_qray + w * 6 * 64;
    qr[l] = ray.o.x; qr[64 + l] = ray.o.y; qr[128 + l] = ray.o.z;
    qr[192 + l] = ray.d.x; qr[256 + l] = ray.d.y; qr[320 + l] = ray.d.z;
    rt_qbound[w * 64 + l] = active && bound > 0. ? (unsigned long long)__double_as_longlong(bound) : 0ull;
    rt_qrank[w * 64 + l] = 0;
    wave_lds_sync();
}
// A distance of either sign as an unsigned word in the same order (the walk of an object tree: a ray that starts
// inside a sphere meets it at a negative distance, sphere.rs:33-62); never 0 for a number.
DEV unsigned long long dkey(double t) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(t);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
DEV double dkey_inv(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}
// Parks the rays for a closest-hit walk of the object tree: the bound word is dkey(best distance so far), 0 = no ray;
// `rank` is the object that holds that distance (a later object must be nearer, or as near with a lower index,
// scene.rs:57-82).
DEV void park_rays_objects(const Ray& ray, bool active, double bound, uint32_t rank) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    double* qr = rt_qray + w * 6 * 64;
    qr[l] = ray.o.x; qr[64 + l] = ray.o.y; qr[128 + l] = ray.o.z;
    qr[192 + l] = ray.d.x; qr[256 + l] = ray.d.y; qr[320 + l] = ray.d.z;
    rt_qbound[w * 64 + l] = active ? dkey(bound) : 0ull;
    rt_qrank[w * 64 + l] = rank;
    wave_lds_sync();
}
DEV bool parked_blocked() { return rt_qbound[uni(threadIdx.x >> 6) * 64 + lane_now()] == 0; }   // (one word per thread of the block)

// One batch: lane k takes pair base + k (k < cnt).
__device__ void test_pairs(const DTri* tris, uint32_t base, uint32_t cnt, Visits& vc) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    const double* qr = rt_qray + w * 6 * 64;
    unsigned long long* qb = rt_qbound + w * 64;
    wave_lds_sync();
    vc.wtris++;
    if (l < cnt) {
        const uint32_t pair = rt_qpair[w * RT_QCAP + base + l];
        uint32_t rank = pair >> 6;
        const uint32_t owner = pair & 63u;
        const unsigned long long b = qb[owner];
        if (b != 0) {
            const D3 rd = d3(qr[192 + owner], qr[256 + owner], qr[320 + owner]);
            while (true) {
                // The origin is read from LDS again for every triangle (most leaves hold one or two): held in registers
                // across the loop it is what the allocator spills to scratch memory, 16 bytes per lane and round.
                asm volatile("" ::: "memory");
                const D3 ro = d3(qr[owner], qr[64 + owner], qr[128 + owner]);
                const TriData T = tri_load(gptr(tris) + rank);
                double t; float u, v;
                vc.tris++; vc.units += 5;
                if (tri_test_od(T, ro, rd, t, u, v) && (unsigned long long)__double_as_longlong(t) < b) { qb[owner] = 0; break; }
                if (T.last) break;
                ++rank;
            }
        }
    }
    wave_lds_sync();
}

// the f32 distance limit of a bound word (never below the bound itself)
template <bool CLOSEST, bool OBJ = false>
DEV float bound_limit(unsigned long long b) {
    if (CLOSEST && OBJ) return limit_up(dkey_inv(b));
    if (CLOSEST) return limit_up(__longlong_as_double((long long)b));
    // b - 1 = the bits of the bound, or 0 (a bound that cannot be met: the box distance limit 0 is as good as any)
    return b > 1 ? limit_up(__longlong_as_double((long long)(b - 1ull))) : 0.f;
}

// One batch of a closest-hit walk: lane k takes pair base + k and finds the best hit among its leaf's triangles that
// beats the owner's best; then the lanes that worked for one owner settle it between them in LDS: the nearest
// distance by an atomic minimum on the bound word (positive doubles order like their bit patterns), among equals
// the lowest triangle position by a second minimum, and the winner files its barycentrics.
__device__ void test_pairs_closest(const DTri* tris, uint32_t base, uint32_t cnt, Visits& vc) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    const double* qr = rt_qray + w * 6 * 64;
    unsigned long long* qb = rt_qbound + w * 64;
    uint32_t* qk = rt_qrank + w * 64;
    wave_lds_sync();
    vc.wtris++;
    bool found = false;
    uint32_t owner = 0, brank = 0;
    unsigned long long b = 0, bt = 0;
    float bu = 0.f, bv = 0.f;
    if (l < cnt) {
        const uint32_t pair = rt_qpair[w * RT_QCAP + base + l];
        uint32_t rank = pair >> 6;
        owner = pair & 63u;
        b = qb[owner];
        if (b != 0) {
            bt = b; brank = qk[owner];
            const D3 rd = d3(qr[192 + owner], qr[256 + owner], qr[320 + owner]);
            while (true) {
                asm volatile("" ::: "memory");
                const D3 ro = d3(qr[owner], qr[64 + owner], qr[128 + owner]);
                const TriData T = tri_load(gptr(tris) + rank);
                double t; float u, v;
                vc.tris++; vc.units += 5;
                if (tri_test_od(T, ro, rd, t, u, v)) {
                    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
                    if (tb < bt || (tb == bt && rank < brank)) { bt = tb; brank = rank; bu = u; bv = v; found = true; }
                }
                if (T.last) break;
                ++rank;
            }
        }
    }
    if (__ballot(found)) {
        if (found) atomicMin(&qb[owner], bt);
        wave_lds_sync();
        const bool cand = found && qb[owner] == bt;
        if (cand && bt < b) qk[owner] = 0xFFFFFFFFu;   // a nearer distance than before the round: the old position is void
        wave_lds_sync();
        if (cand) atomicMin(&qk[owner], brank);
        wave_lds_sync();
        if (cand && qk[owner] == brank) rt_quv[w * 64 + owner] = make_float2(bu, bv);
    }
    wave_lds_sync();
}

// One batch of the walk of an object tree: lane k takes pair base + k = (object, ray) and runs the object's own test
// (sphere.rs:33-62, aabb.rs:18-67).  Shadow walk: a hit no farther than the bound clears the owner's bound word.
// Closest walk: as test_pairs_closest, with the object's index in the place of the triangle's position.
template <bool CLOSEST>
__device__ void test_pairs_obj(const DScene& S, uint32_t base, uint32_t cnt, Visits& vc) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    const double* qr = rt_qray + w * 6 * 64;
    unsigned long long* qb = rt_qbound + w * 64;
    uint32_t* qk = rt_qrank + w * 64;
    wave_lds_sync();
    vc.wtris++;
    bool found = false;
    uint32_t owner = 0, oi = 0;
    unsigned long long b = 0, key = 0;
    if (l < cnt) {
        const uint32_t pair = rt_qpair[w * RT_QCAP + base + l];
        oi = pair >> 6;
        owner = pair & 63u;
        b = qb[owner];
        if (b != 0) {
            Ray r;
            r.o = d3(qr[owner], qr[64 + owner], qr[128 + owner]);
            r.d = d3(qr[192 + owner], qr[256 + owner], qr[320 + owner]);
            const RT_GLOBAL DObject* op = gptr(S.objects) + oi;
            const uint32_t kind = op->kind, payload = op->payload;
            double near; bool hit;
            vc.tris++; vc.units += 3;
            if (kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)payload, r, near);
            else { r.m = d3(1. / r.d.x, 1. / r.d.y, 1. / r.d.z); hit = box_near(gptr(S.boxes) + 6 * (size_t)payload, r, near); }
            if (hit) {
                if (!CLOSEST) { if (b > 1 && near <= __longlong_as_double((long long)(b - 1ull))) qb[owner] = 0; }
                else { key = dkey(near); found = key < b || (key == b && oi < qk[owner]); }
            }
        }
    }
    if (CLOSEST && __ballot(found)) {
        if (found) atomicMin(&qb[owner], key);
        wave_lds_sync();
        const bool cand = found && qb[owner] == key;
        if (cand && key < b) qk[owner] = 0xFFFFFFFFu;
        wave_lds_sync();
        if (cand) atomicMin(&qk[owner], oi);
    }
    wave_lds_sync();
}

// The thin ends of the walk.  Near the leaves the lanes of a packet have gone separate ways: half of a terrain's node
// steps serve fewer than 16 of the 64 lanes, a quarter fewer than 4, and a step costs the wave the same whatever the
// number.  So a child that few lanes entered is not walked by the packet: its (lane, node) pairs wait in a second
// queue, and a round of node tests gives every lane of the wave one pair -- that ray from LDS, that node's record
// by vector loads, the same box arithmetic (entry and exit found per slab by min/max instead of by the octant, to
// the same values) -- and files the children entered as new pairs: leaves with the triangle pairs, inner nodes back
// here.  The newest pairs go first (depth first), which keeps the queue short.
// p = pairs to take (from the end of the queue); pn, qn = fill of the node and the triangle queue.
// CLOSEST: a pair carries its node's entry distance and is dropped unread when the ray's best hit has come nearer
// since; each lane files the children it entered farthest first, so that its nearest is the first to be taken.
// OBJ: the tree is a scene's object tree (distances of either sign: nothing is clamped at the ray's origin).
template <bool CLOSEST, bool OBJ>
__device__ void node_pairs(const DNode* nodes, float coord_max, uint32_t p, uint32_t& pn, uint32_t& qn, Visits& vc) {
    const uint32_t w = uni(threadIdx.x >> 6), l = lane_now();
    const uint32_t wq = w * RT_QCAP, wn = w * RT_NCAP;
    wave_lds_sync();
    pn -= p;
    const float INF = asf(0x7F800000u);
    bool h0 = false, h1 = false, h2 = false, h3 = false;
    float t0 = INF, t1 = INF, t2 = INF, t3 = INF;
    uint32_t owner = 0;
    rt_u4 ref; ref.x = ref.y = ref.z = ref.w = RT_NO_CHILD;
    vc.wnodes++;
    if (l < p) {
        const uint32_t e = rt_qnode[wn + pn + l];
        owner = e & 63u;
        const unsigned long long b = rt_qbound[w * 64 + owner];
        const float limit = bound_limit<CLOSEST, OBJ>(b);
        bool go = b != 0;
        if (CLOSEST) go = go && !(rt_qntn[wn + pn + l] > limit);
        if (go) {
            const double* qr = rt_qray + w * 6 * 64 + owner;
            PRay pr;
            pray_axis_d(qr[0], qr[192], coord_max, pr.mx, pr.cminx, pr.cmaxx);
            pray_axis_d(qr[64], qr[256], coord_max, pr.my, pr.cminy, pr.cmaxy);
            pray_axis_d(qr[128], qr[320], coord_max, pr.mz, pr.cminz, pr.cmaxz);
            const RT_GLOBAL rt_u4* rec = (const RT_GLOBAL rt_u4*)(gptr(nodes) + (e >> 6));
            UNode n;
            n.minx = rec[0]; n.miny = rec[1]; n.minz = rec[2]; n.maxx = rec[3]; n.maxy = rec[4]; n.maxz = rec[5]; ref = rec[6];
            vc.nodes++; vc.units += 7;
            float f0, f1, f2, f3;
            RT_CHILD_PAIR(n, 0, t0, f0, t1, f1);
            RT_CHILD_PAIR(n, 2, t2, f2, t3, f3);
            if (CLOSEST && OBJ) {
                h0 = t0 <= fminf(f0, limit) && f0 >= 0.f;
                h1 = t1 <= fminf(f1, limit) && f1 >= 0.f;
                h2 = t2 <= fminf(f2, limit) && f2 >= 0.f && ref.z != RT_NO_CHILD;
                h3 = t3 <= fminf(f3, limit) && f3 >= 0.f && ref.w != RT_NO_CHILD;
            } else {
                h0 = fmaxf(t0, 0.f) <= fminf(f0, limit);
                h1 = fmaxf(t1, 0.f) <= fminf(f1, limit);
                h2 = fmaxf(t2, 0.f) <= fminf(f2, limit) && ref.z != RT_NO_CHILD;
                h3 = fmaxf(t3, 0.f) <= fminf(f3, limit) && ref.w != RT_NO_CHILD;
            }
        }
    }
#define RT_PAIR_CHILD(h_, r_, t_)                                                                          \
    {                                                                                                      \
        const unsigned long long mh_ = __ballot(h_), leaf_ = __ballot((int)(r_) < 0);                      \
        const unsigned long long ml_ = mh_ & leaf_, mi_ = mh_ & ~leaf_;                                    \
        if (in_mask(ml_)) rt_qpair[wq + qn + lanes_below(ml_)] = (((r_) & ~RT_LEAF_BIT) << 6) | owner;     \
        qn += (uint32_t)__popcll(ml_);                                                                     \
        if (in_mask(mi_)) {                                                                                \
            const uint32_t at_ = wn + pn + lanes_below(mi_);                                               \
            rt_qnode[at_] = ((r_) << 6) | owner;                                                           \
            if (CLOSEST) rt_qntn[at_] = (t_);                                                              \
        }                                                                                                  \
        pn += (uint32_t)__popcll(mi_);                                                                     \
    }
    if (CLOSEST) {
        // this lane's children by entry distance (missed ones last), then four passes, farthest first
        uint32_t r0 = ref.x, r1 = ref.y, r2 = ref.z, r3 = ref.w;
        t0 = h0 ? t0 : INF; t1 = h1 ? t1 : INF; t2 = h2 ? t2 : INF; t3 = h3 ? t3 : INF;
        RT_CSWAP(t0, r0, t1, r1); RT_CSWAP(t2, r2, t3, r3);
        RT_CSWAP(t0, r0, t2, r2); RT_CSWAP(t1, r1, t3, r3);
        RT_CSWAP(t1, r1, t2, r2);
        RT_PAIR_CHILD(t3 < INF, r3, t3)
        RT_PAIR_CHILD(t2 < INF, r2, t2)
        RT_PAIR_CHILD(t1 < INF, r1, t1)
        RT_PAIR_CHILD(t0 < INF, r0, t0)
    } else {
        RT_PAIR_CHILD(h3, ref.w, t3)
        RT_PAIR_CHILD(h2, ref.z, t2)
        RT_PAIR_CHILD(h1, ref.y, t1)
        RT_PAIR_CHILD(h0, ref.x, t0)
    }
#undef RT_PAIR_CHILD
    wave_lds_sync();
}

// The walk of one mesh for the parked rays.  Every lane of the wave must call it.
// CLOSEST = false, the shadow walk: lanes found blocked have their bound word cleared.  With BUDGET the wave gives
// up after `budget` inner nodes and reports the lanes that were still open in `unresolved` (their rays are then
// finished piecewise by other lanes, mesh_traverse_seg); BUDGET = false compiles the step count out (large launches
// run without a budget).
// CLOSEST = true: every ray's nearest hit (bound word, rt_qrank, rt_quv).  Nobody leaves the packet early, but the
// distance limit of the box tests follows each ray's best hit, and children are taken nearest first.
// OBJ: Mp describes the scene's object tree (DScene::tlas_mesh: same records, a leaf child is an object), `Sp` is the
// scene; the parked rays are those whose direction is of unit length (the others do not cull, tlas_culls).
template <bool BUDGET, bool CLOSEST, bool OBJ = false>
__device__ void packet_deferred(const DMesh* Mp, Visits& vc, uint32_t budget, unsigned long long& unresolved, const DScene* Sp = nullptr) {
    unresolved = 0;
    uint32_t steps = 0;
    // the mesh record is the same for the whole wave: read through the scalar cache, so that neither its address nor
    // its fields occupy vector registers across the walk
    const RT_CONST DMesh& Mg = *(const RT_CONST DMesh*)uni_ptr(Mp);
    const DNode* nodes = Mg.nodes;
    const DTri* tris = Mg.tris;
    const uint32_t w = uni(threadIdx.x >> 6);
    const uint32_t wq = w * RT_QCAP, wn = w * RT_NCAP;
    const uint32_t ncap = uni(rt_node_cap) < RT_NCAP ? uni(rt_node_cap) : RT_NCAP;
    const uint32_t qflush = CLOSEST && OBJ ? RT_QFLUSH_OBJ : RT_QFLUSH;
    // Per-lane constants of the box test, as in PRay but filed by role: c_entry goes with the plane the ray meets
    // first along the axis, c_exit with the other.  Which plane that is (min or max) depends on the direction's sign
    // only, and the walk below handles lanes of one sign pattern at a time, so it is settled per walk, not per test.
    // They are worked out from the parked ray at the start of every stretch of the walk and belong to that stretch
    // alone (a scope of their own): a round of triangle tests wants every register it can get, and a value that is
    // live on ANY path across the round is kept, or spilled to scratch memory, on all of them.
    struct BoxK { float mx, my, mz, enx, eny, enz, exx, exy, exz, limit; };
    auto box_constants = [&]() {
        BoxK k;
        asm volatile("" ::: "memory");   // (the parked ray is read again, not remembered from the last stretch)
        const uint32_t l_ = lane_now();
        const double* qr_ = rt_qray + w * 6 * 64 + l_;
        const unsigned long long b_ = rt_qbound[w * 64 + l_];
        const float B_ = Mg.coord_max;
        float lo_, hi_;
        pray_axis_d(qr_[0], qr_[192], B_, k.mx, lo_, hi_); k.enx = fminf(lo_, hi_); k.exx = fmaxf(lo_, hi_);
        pray_axis_d(qr_[64], qr_[256], B_, k.my, lo_, hi_); k.eny = fminf(lo_, hi_); k.exy = fmaxf(lo_, hi_);
        pray_axis_d(qr_[128], qr_[320], B_, k.mz, lo_, hi_); k.enz = fminf(lo_, hi_); k.exz = fmaxf(lo_, hi_);
        k.limit = bound_limit<CLOSEST, OBJ>(b_);
        return k;
    };
    unsigned long long remaining;
    uint32_t my_octant;
    {
        // the root's own box, with the same constants
        const BoxK k = box_constants();
        const RT_CONST float* lo = Mg.root_min; const RT_CONST float* hi = Mg.root_max;
        const bool nx = k.mx < 0.f, ny = k.my < 0.f, nz = k.mz < 0.f;
        my_octant = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
        const float ax = fmaf(nx ? hi[0] : lo[0], k.mx, k.enx), bx = fmaf(nx ? lo[0] : hi[0], k.mx, k.exx);
        const float ay = fmaf(ny ? hi[1] : lo[1], k.my, k.eny), by = fmaf(ny ? lo[1] : hi[1], k.my, k.exy);
        const float az = fmaf(nz ? hi[2] : lo[2], k.mz, k.enz), bz = fmaf(nz ? lo[2] : hi[2], k.mz, k.exz);
        const float tn = fmaxf(fmaxf(ax, ay), az), tf = fminf(fminf(bx, by), bz);
        const bool root_hit = rt_qbound[w * 64 + lane_now()] != 0 &&
                              (CLOSEST && OBJ ? tn <= fminf(tf, k.limit) && tf >= 0.f : fmaxf(tn, 0.f) <= fminf(tf, k.limit));
        remaining = __ballot(root_hit);
    }
    if (!remaining) return;
    const uint32_t root = Mg.root_ref;
    uint32_t qn = 0, pn = 0;
    RStack st; st.ref = 0; st.lo = 0; st.hi = 0; st.sp = 0;
    unsigned long long alive = 0;

    // a child some lane entered: a leaf's (lane, leaf) pairs go on the triangle queue, an inner node on the stack,
    // or, entered by few lanes (and while there is room), on the node queue
#define RT_CHILD_ENTERED(r_, m_, t_)                                                                \
    if (m_) {                                                                                       \
        const uint32_t c_ = (uint32_t)__popcll(m_);                                                 \
        if ((r_) & RT_LEAF_BIT) {                                                                   \
            if (in_mask(m_)) rt_qpair[wq + qn + lanes_below(m_)] = (((r_) & ~RT_LEAF_BIT) << 6) | lane_now(); \
            qn += c_;                                                                               \
        } else if (c_ <= (CLOSEST ? RT_SPARSE_CLOSEST : RT_SPARSE) && pn + c_ <= ncap) {            \
            if (in_mask(m_)) {                                                                      \
                const uint32_t at_ = wn + pn + lanes_below(m_);                                     \
                rt_qnode[at_] = ((r_) << 6) | lane_now();                                           \
                if (CLOSEST) rt_qntn[at_] = (t_);                                                   \
            }                                                                                       \
            pn += c_;                                                                               \
        } else st.push((r_), (m_));                                                                 \
    }
    // children k and k+1: entry distance = the latest of the three entry planes, exit = the earliest exit plane
#define RT_PAIR_ENTRY_EXIT(k, tn_a, tf_a, tn_b, tf_b)                                                                     \
    do {                                                                                                                  \
        const rt_f2 ax_ = __builtin_elementwise_fma(rt_f2{asf(pen_x[k]), asf(pen_x[k + 1])}, rt_f2{K.mx, K.mx}, rt_f2{K.enx, K.enx}); \
        const rt_f2 bx_ = __builtin_elementwise_fma(rt_f2{asf(pex_x[k]), asf(pex_x[k + 1])}, rt_f2{K.mx, K.mx}, rt_f2{K.exx, K.exx}); \
        const rt_f2 ay_ = __builtin_elementwise_fma(rt_f2{asf(pen_y[k]), asf(pen_y[k + 1])}, rt_f2{K.my, K.my}, rt_f2{K.eny, K.eny}); \
        const rt_f2 by_ = __builtin_elementwise_fma(rt_f2{asf(pex_y[k]), asf(pex_y[k + 1])}, rt_f2{K.my, K.my}, rt_f2{K.exy, K.exy}); \
        const rt_f2 az_ = __builtin_elementwise_fma(rt_f2{asf(pen_z[k]), asf(pen_z[k + 1])}, rt_f2{K.mz, K.mz}, rt_f2{K.enz, K.enz}); \
        const rt_f2 bz_ = __builtin_elementwise_fma(rt_f2{asf(pex_z[k]), asf(pex_z[k + 1])}, rt_f2{K.mz, K.mz}, rt_f2{K.exz, K.exz}); \
        (tn_a) = fmaxf(fmaxf(ax_.x, ay_.x), az_.x); (tf_a) = fminf(fminf(bx_.x, by_.x), bz_.x);                            \
        (tn_b) = fmaxf(fmaxf(ax_.y, ay_.y), az_.y); (tf_b) = fminf(fminf(bx_.y, by_.y), bz_.y);                            \
    } while (0)

    // The lanes are walked in groups of one direction octant each (shadow rays towards one light nearly always share
    // it, so there is one group).  Within a group a node record's planes are fetched as "entry planes" and "exit
    // planes" -- the octant picks the offsets of the six loads -- and a box costs 3 packed FMAs per two children
    // plus max3, min3, two clamps and a compare: no sorting of each slab's two distances.
    while (remaining) {
        const uint32_t octant = (uint32_t)__builtin_amdgcn_readlane((int)my_octant, __ffsll((long long)remaining) - 1);
        alive = __ballot(my_octant == octant) & remaining;
        remaining &= ~alive;
        // 16-byte words of a node record: min x, y, z = 0, 1, 2; max x, y, z = 3, 4, 5; references = 6
        const uint32_t i_en_x = (octant & 1u) ? 3u : 0u, i_ex_x = (octant & 1u) ? 0u : 3u;
        const uint32_t i_en_y = (octant & 2u) ? 4u : 1u, i_ex_y = (octant & 2u) ? 1u : 4u;
        const uint32_t i_en_z = (octant & 4u) ? 5u : 2u, i_ex_z = (octant & 4u) ? 2u : 5u;
        qn = 0; pn = 0;   // (pairs left over from a group whose lanes are all blocked)
        st.sp = 0;
        if (root & RT_LEAF_BIT) {
            // a mesh of one leaf: nothing to walk, every lane of the group takes the leaf to the triangle rounds
            if (in_mask(alive)) rt_qpair[wq + lanes_below(alive)] = ((root & ~RT_LEAF_BIT) << 6) | lane_now();
            qn = (uint32_t)__popcll(alive);
        } else st.push(root, alive);
        while (true) {
            if (qn >= qflush || (qn > 0 && st.sp == 0 && pn == 0)) {
                // one round of triangle tests (the one place where triangles are tested, so that the kernel holds that
                // code once): when 48 pairs wait, or nothing else is left to do
                const uint32_t cnt = qn < 64 ? qn : 64;
                qn -= cnt;
                if (OBJ) test_pairs_obj<CLOSEST>(*Sp, qn, cnt, vc);
                else if (CLOSEST) test_pairs_closest(tris, qn, cnt, vc);
                else test_pairs(tris, qn, cnt, vc);
                if (!CLOSEST) {
                    alive &= ~__ballot(parked_blocked());
                    if (!alive) break;
                }
                continue;
            }
            if (pn >= RT_NFLUSH || (pn > 0 && st.sp == 0)) {
                // one round of node tests: when 64 pairs wait, or the packet has nothing left to walk.  A round may
                // add three pairs for every one it takes, so a queue that is nearly full takes fewer; a full one hands
                // its newest pair back to the packet walk (whose stack is bounded by the tree's depth).
                if (BUDGET && ++steps > budget) { unresolved = 1; break; }
                uint32_t p = pn < 64 ? pn : 64;
                const uint32_t room = (ncap - pn) / 3;
                if (room < p) p = room;
                if (p == 0) {
                    const uint32_t e = uni(rt_qnode[wn + pn - 1]);
                    --pn;
                    st.push(e >> 6, 1ull << (e & 63u));
                } else {
                    node_pairs<CLOSEST, OBJ>(nodes, Mg.coord_max, p, pn, qn, vc);
                    continue;
                }
            }
            if (st.sp == 0) break;   // (and both queues are empty)
            // one stretch of the packet walk: until a queue wants a round or nothing is left to walk
            {
                const BoxK K = box_constants();
                do {
                    const unsigned long long inmask = st.top_mask() & alive;
                    const uint32_t cur = st.top_ref();
                    --st.sp;
                    if (!inmask) continue;   // everybody who wanted this node has been blocked since
                    if (BUDGET && ++steps > budget) { unresolved = 1; st.sp = 0; qn = 0; pn = 0; break; }
                    const RT_CONST rt_u4* rec = (const RT_CONST rt_u4*)(nodes + cur);
                    const rt_u4 pen_x = rec[i_en_x], pex_x = rec[i_ex_x], pen_y = rec[i_en_y], pex_y = rec[i_ex_y];
                    const rt_u4 pen_z = rec[i_en_z], pex_z = rec[i_ex_z], ref = rec[6];
                    if (in_mask(inmask)) vc.nodes++;
                    vc.wnodes++;
                    if (__lane_id() == 0) vc.units += 7;
                    float t0, t1, t2, t3, f0, f1, f2, f3;
                    RT_PAIR_ENTRY_EXIT(0, t0, f0, t1, f1);
                    RT_PAIR_ENTRY_EXIT(2, t2, f2, t3, f3);
                    // the loop runs with every lane of the wave enabled, so a compare's lane mask is already the ballot
                    const unsigned long long m0 = box_lanes<CLOSEST && OBJ>(t0, f0, K.limit) & inmask, m1 = box_lanes<CLOSEST && OBJ>(t1, f1, K.limit) & inmask;
                    unsigned long long m2 = box_lanes<CLOSEST && OBJ>(t2, f2, K.limit) & inmask, m3 = box_lanes<CLOSEST && OBJ>(t3, f3, K.limit) & inmask;
                    if (ref[2] == RT_NO_CHILD) m2 = 0;
                    if (ref[3] == RT_NO_CHILD) m3 = 0;
                    if (CLOSEST) {
                        // nearest first, by the entry distance of the first lane in each child (rays of a packet
                        // agree mostly); children nobody entered sort last
                        const float INF = asf(0x7F800000u);
                        unsigned long long s0 = m0, s1 = m1, s2 = m2, s3 = m3;
                        uint32_t r0 = ref[0], r1 = ref[1], r2 = ref[2], r3 = ref[3];
                        float k0 = s0 ? first_lane_f(t0, s0) : INF, k1 = s1 ? first_lane_f(t1, s1) : INF;
                        float k2 = s2 ? first_lane_f(t2, s2) : INF, k3 = s3 ? first_lane_f(t3, s3) : INF;
                        RT_UCSWAP_T(k0, r0, s0, t0, k1, r1, s1, t1); RT_UCSWAP_T(k2, r2, s2, t2, k3, r3, s3, t3);
                        RT_UCSWAP_T(k0, r0, s0, t0, k2, r2, s2, t2); RT_UCSWAP_T(k1, r1, s1, t1, k3, r3, s3, t3);
                        RT_UCSWAP_T(k1, r1, s1, t1, k2, r2, s2, t2);
                        RT_CHILD_ENTERED(r3, s3, t3)
                        RT_CHILD_ENTERED(r2, s2, t2)
                        RT_CHILD_ENTERED(r1, s1, t1)
                        RT_CHILD_ENTERED(r0, s0, t0)
                    } else {
                        RT_CHILD_ENTERED(ref[3], m3, t3)
                        RT_CHILD_ENTERED(ref[2], m2, t2)
                        RT_CHILD_ENTERED(ref[1], m1, t1)
                        RT_CHILD_ENTERED(ref[0], m0, t0)   // pushed last: walked first (the order makes no measurable difference)
                    }
                } while (st.sp > 0 && qn < qflush && pn < RT_NFLUSH);
            }
        }
        if (BUDGET && unresolved) { unresolved = alive | remaining; break; }   // out of budget: the lanes still open
    }
#undef RT_PAIR_ENTRY_EXIT
#undef RT_CHILD_ENTERED
}
template <bool BUDGET>
DEV void packet_any_deferred(const DMesh* Mp, Visits& vc, uint32_t budget, unsigned long long& unresolved) {
    packet_deferred<BUDGET, false>(Mp, vc, budget, unresolved);
}
// Closest hit of the wave's rays in one mesh, nearer than `bound` (ties to the lower leaf-order position): the walk
// above with the rays parked for it.  Every lane of the wave must call it.
__device__ bool packet_closest_deferred(const DMesh* Mp, const Ray& ray, bool active, double bound, MeshHit& mh, Visits& vc) {
    park_rays_closest(ray, active, bound);
    unsigned long long unresolved;
    packet_deferred<false, true>(Mp, vc, 0, unresolved);
    wave_lds_sync();
    const uint32_t at = uni(threadIdx.x >> 6) * 64 + lane_now();
    const unsigned long long b = rt_qbound[at];
    mh.n = 0; mh.t0 = bound; mh.t1 = bound; mh.r0 = 0; mh.u = 0.f; mh.v = 0.f;
    if (active && b != 0 && b < (unsigned long long)__double_as_longlong(bound)) {
        mh.n = 1; mh.t0 = __longlong_as_double((long long)b); mh.t1 = mh.t0; mh.r0 = rt_qrank[at];
        const float2 uv = rt_quv[at]; mh.u = uv.x; mh.v = uv.y;
    }
    return mh.n > 0;
}

// Closest object of the scene's object tree for the wave's rays: nearer than `bound`, or as near with an index below
// `rank` (scene.rs:57-82: the stable sort keeps the lowest index among equals).  Every lane of the wave must call it.
__device__ bool packet_closest_objects(const DScene* Sp, const Ray& ray, bool active, double bound, uint32_t rank, double& near, uint32_t& oi, Visits& vc) {
    park_rays_objects(ray, active, bound, rank);
    unsigned long long unresolved;
    packet_deferred<false, true, true>(&Sp->tlas_mesh, vc, 0, unresolved, Sp);
    wave_lds_sync();
    const uint32_t at = uni(threadIdx.x >> 6) * 64 + lane_now();
    const unsigned long long b = rt_qbound[at], b0 = dkey(bound);
    const uint32_t k = rt_qrank[at];
    if (active && b != 0 && (b < b0 || (b == b0 && k < rank))) { near = dkey_inv(b); oi = k; return true; }
    return false;
}

// Scene::cast_ray_once for a whole wave (scene.rs:57-82): the object loop is uniform, meshes are walked as a
// packet.  Every lane calls it; `active` lanes carry a ray.
// FAR = false: the scene has no mesh whose material needs Hit::far, and the lane-by-lane two-hit walk is left out
// of the kernel (it costs registers, hence waves per SIMD, even when it never runs).
// DEFER: the meshes are walked by packet_closest_deferred (triangle tests and thin ends of the walk set aside)
// instead of packet_traverse.
// RECIP = false: the caller vouches that no cube is on the linear list and that the scene has no object tree; `ray`
// then needs its origin and direction only.
template <bool FAR, bool DEFER = false, bool RECIP = true>
__device__ bool cast_closest_packet(const DScene* Sp, const Ray& ray, bool active, Closest& c, Visits& vc) {
    const DScene& S = *Sp;
    c.obj = RT_NONE; c.rank = RT_NONE; c.t = 0.; c.t2 = 0.; c.u = 0.f; c.v = 0.f;
    bool any = false;
    const uint32_t n = S.n_linear;
    const RT_GLOBAL DObject* objs = gptr(S.objects);
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = gptr(S.linear)[li_];   // the objects outside the object tree, in scene order
        const uint32_t kind = uni(objs[oi].kind), payload = uni(objs[oi].payload), need_far = uni(objs[oi].need_far);
        double near = 0.; bool hit = false;
        if (kind == 3) {
            MeshHit mh;
            if (FAR && need_far) {
                if (active && mesh_traverse<TRAV_TWO>(S.meshes[payload], ray, INFINITY, mh, vc) && (!any || mh.t0 < c.t)) {
                    c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.n >= 2 ? mh.t1 : mh.t0; c.u = mh.u; c.v = mh.v; any = true;
                }
            } else if (DEFER ? packet_closest_deferred(S.meshes + payload, ray, active, any ? c.t : INFINITY, mh, vc)
                             : packet_traverse(S.meshes[payload], ray, active, any ? c.t : INFINITY, mh, vc)) {
                c.obj = oi; c.rank = mh.r0; c.t = mh.t0; c.t2 = mh.t0; c.u = mh.u; c.v = mh.v; any = true;
            }
            continue;
        }
        if (active) {
            if (kind == 0) hit = sphere_near(gptr(S.spheres) + 4 * (size_t)payload, ray, near);
            else if (kind == 1) hit = plane_near(gptr(S.planes) + 8 * (size_t)payload, ray, near);
            else if (RECIP) hit = box_near(gptr(S.boxes) + 6 * (size_t)payload, ray, near);
        }
        if (hit && (!any || near < c.t)) { c.obj = oi; c.rank = RT_NONE; c.t = near; any = true; }
    }
    if (RECIP && S.n_tlas) {
        if (DEFER) {
            // rays whose direction is not of unit length do not cull (tlas_culls): lane by lane, every object
            const bool culls = tlas_culls(ray);
            if (active && !culls) any = tlas_closest(S, ray, c, any);
            double near; uint32_t oi;
            if (packet_closest_objects(Sp, ray, active && culls, any ? c.t : INFINITY, any ? c.obj : RT_NONE, near, oi, vc)) {
                c.obj = oi; c.rank = RT_NONE; c.t = near; c.t2 = 0.; c.u = 0.f; c.v = 0.f; any = true;
            }
        } else if (active) any = tlas_closest(S, ray, c, any);
    }
    return any;
}

// Shadow cast for a whole wave (sun.rs:89, point.rs:91-92).  "Blocked" is an OR over the objects, so their order is
// free: the analytic objects go first (cheap, and every ray they block leaves the packets), the meshes last (after
// the last walk nothing of the ray is needed any more, so it does not have to survive the walk in registers).
// With a budget, a wave whose mesh walk runs long gives up: `unresolved` then holds the lanes whose cast is still
// open, to be cast again piecewise.
// RECIP = false: the caller vouches that no cube is on the linear list and that the scene has no object tree; `ray`
// then needs its origin and direction only.
template <bool BUDGET, bool RECIP = true>
__device__ bool cast_any_packet(const DScene* Sp, const Ray& ray, bool active, bool limited, double dist, Visits& vc,
                                uint32_t budget, unsigned long long& unresolved) {
    const DScene& S = *Sp;
    const uint32_t n = S.n_linear;
    // the object list and the analytic objects' records are the same for the whole wave and never written while a frame
    // renders: read through the scalar cache (constant address space), they cost neither vector loads nor registers
    // (the closest-hit packets, shorter of scalar registers, keep their vector loads)
    const RT_CONST DObject* objs = (const RT_CONST DObject*)uni_ptr(S.objects);
    const RT_CONST uint32_t* linear = (const RT_CONST uint32_t*)uni_ptr(S.linear);
    const RT_CONST double* spheres = (const RT_CONST double*)uni_ptr(S.spheres);
    const RT_CONST double* planes = (const RT_CONST double*)uni_ptr(S.planes);
    bool blocked = false;
    unresolved = 0;
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = linear[li_];   // the objects outside the object tree, in scene order
        const uint32_t kind = objs[oi].kind, payload = objs[oi].payload;
        if (kind == 3) continue;
        if (active && !blocked) {
            double near; bool hit;
            if (kind == 0) hit = sphere_near(spheres + 4 * (size_t)payload, ray, near);
            else if (kind == 1) hit = plane_near(planes + 8 * (size_t)payload, ray, near);
            else if (RECIP) hit = box_near(gptr(S.boxes) + 6 * (size_t)payload, ray, near);
            else hit = false;
            if (hit && (!limited || near <= dist)) blocked = true;
        }
    }
    // the object tree: as a packet too, but for rays whose direction is not of unit length (they do not cull, tlas_culls)
    if (RECIP && S.n_tlas && active && !blocked && !tlas_culls(ray)) blocked = tlas_any(S, ray, limited, dist);
    if (!__ballot(active && !blocked)) return blocked;
    const bool open = active && !blocked;   // the lanes whose cast the object tree and the meshes decide
    park_rays(ray, open, limited ? dist : INFINITY);
    if (RECIP && S.n_tlas) {
        unsigned long long unres = 0;
        packet_deferred<BUDGET, false, true>(&Sp->tlas_mesh, vc, budget, unres, Sp);
        const unsigned long long still = __ballot(open && !parked_blocked());
        if (unres) { unresolved = still; return blocked || (open && parked_blocked()); }
        if (!still) return blocked || (open && parked_blocked());
    }
    for (uint32_t li_ = 0; li_ < n; li_++) {
        const uint32_t oi = linear[li_];   // the objects outside the object tree, in scene order
        const uint32_t kind = objs[oi].kind, payload = objs[oi].payload;
        if (kind != 3) continue;
        unsigned long long unres = 0;
        packet_any_deferred<BUDGET>(S.meshes + payload, vc, budget, unres);
        const unsigned long long still = __ballot(open && !parked_blocked());
        // every lane that is not blocked yet is cast again: the meshes after this one were not tried for any of them
        if (unres) { unresolved = still; break; }
        if (!still) break;
    }
    return blocked || (open && parked_blocked());
}
