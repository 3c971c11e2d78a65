// Device-side construction and refitting of a mesh's four-wide BVH (bvh_build.hip).
//
// The reference builds a mesh's tree on the host (Mesh::generate_sbvh, mesh.rs:257-272; binned SAH,
// acceleration.rs:296-405) and that stays the default: a caller hands the tree over in rt_mesh::nodes.  For meshes that
// change from frame to frame the host build (0.35 s for a million triangles) dwarfs the frame, so a caller may leave
// rt_mesh::nodes NULL and have the tree built here, on the device, in a few milliseconds; and a mesh whose vertices
// moved but whose connectivity did not can keep its tree and have the boxes refitted.
//
// What a ray hits does not depend on the tree: every triangle test is the reference's exact f64 sequence, and boxes only
// decide which triangles get tested.  The one thing a different tree can change is the winner between two triangles hit
// at EXACTLY the same distance (the reference keeps the candidate met first in its own tree's depth-first order,
// mesh.rs:326-376): here the lower position in THIS tree's leaf order wins.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_types.h"

struct BvhBuildResult {
    DNode* nodes = nullptr;        // hipMalloc'd, n_inner records (nullptr when the mesh is a single triangle)
    DTri* tris = nullptr;          // hipMalloc'd, n_tris records in leaf order
    uint32_t n_inner = 0;
    uint32_t root_ref = 0;
    uint32_t stack_need = 0;       // entries a walk of this tree can have stacked at most
    double root_box[6] = {0, 0, 0, 0, 0, 0};   // min xyz, max xyz of the whole mesh
    double ms_sort = 0., ms_total = 0.;
};

// Builds the tree of an indexed triangle set that already lies in device memory (d_verts: [n_verts][3] f64, d_tri_pos:
// [n_tris][3] u32, indices validated by the caller).  Synchronises the stream.  hipError_t of the first failing call.
hipError_t bvh_build_device(hipStream_t stream, const double* d_verts, const uint32_t* d_tri_pos, uint32_t n_tris, BvhBuildResult* out);

// Recomputes every triangle record and every box of an existing tree from new vertex positions (same connectivity).
// d_tris' `orig` fields say which triangle each leaf-order position holds; nodes' pad words hold the parent links
// (pad[0] parent record or RT_NONE, pad[1] slot in the parent, pad[2] number of inner children).  n_records: how many
// triangle records d_tris holds -- one per leaf entry, which is the triangle count only for a tree that lists every
// triangle exactly once (a caller's tree may list one twice, or leave one out).  Synchronises the stream.
hipError_t bvh_refit_device(hipStream_t stream, const double* d_verts, const uint32_t* d_tri_pos, uint32_t n_records, DNode* nodes, uint32_t n_inner,
                            uint32_t root_ref, DTri* tris, double root_box_out[6]);
