// What rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of the render kernels, against byte counts known
// in advance.  Every kernel below reads every byte of a buffer larger than the 256 MiB on-die cache exactly once, in its
// own pattern; the program prints the bytes each one read, and FETCH_SIZE of the same launches comes from
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o p -- raytracer_amd/fetch_calib.bin
// (tools/fetch_calib_report.py puts the two side by side).
//   stream16   16 bytes per lane, coalesced          (the pattern the counter is known to tally at half)
//   stream8    8 bytes per lane, coalesced            (the f64 planes of a wavefront level)
//   lane128    every lane its own 128-byte record     (a lane-by-lane walk's node fetch)
//   lane80     every lane its own 80-byte record      (a queued triangle test's fetch)
//   wave128    one 128-byte record per wave, fetched through the scalar cache (a packet walk's node fetch)
// Records are visited in a scrambled order (index * odd constant modulo a power of two: a bijection), so that
// neighbouring lanes touch unrelated cache lines, as rays in different parts of a tree do.
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>

#define CONST_AS __attribute__((address_space(4)))

__device__ inline void keep(uint4 acc, uint4* sink) { if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = acc; }   // (never true for a zeroed buffer)
__device__ inline void mix(uint4& a, uint4 v) { a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }

__global__ void stream16(const uint4* p, size_t n_u4, uint4* sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_u4; i += (size_t)gridDim.x * blockDim.x) mix(acc, p[i]);
    keep(acc, sink);
}
__global__ void stream8(const uint2* p, size_t n_u2, uint4* sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_u2; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; }
    keep(acc, sink);
}
// n_rec is a power of two and the launch has exactly n_rec threads: thread t reads record (t * odd) mod n_rec, whole
template <uint32_t REC_U4>
__global__ void lane_records(const uint4* p, uint32_t n_rec, uint4* sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rec) return;
    const uint32_t r = (t * 2654435761u) & (n_rec - 1u);
    const uint4* q = p + (size_t)r * REC_U4;
    uint4 acc = {0, 0, 0, 0};
    #pragma unroll
    for (uint32_t k = 0; k < REC_U4; k++) mix(acc, q[k]);
    keep(acc, sink);
}
// one record per wave: the launch has exactly n_rec waves
__global__ void wave_records(const uint4* p, uint32_t n_rec, uint4* sink) {
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n_rec) return;
    const uint32_t r = (w * 2654435761u) & (n_rec - 1u);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const CONST_AS u4* q = (const CONST_AS u4*)(p + (size_t)r * 8u);
    uint4 acc = {0, 0, 0, 0};
    #pragma unroll
    for (uint32_t k = 0; k < 8; k++) { const u4 v = q[k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    keep(acc, sink);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("{\"error\": \"%s\"}\n", hipGetErrorString(e_)); return 2; } } while (0)

int main() {
    const uint32_t N_REC = 1u << 24;                       // records per gather kernel
    const size_t BYTES = (size_t)N_REC * 128;              // 2 GiB
    void* buf = nullptr; uint4* sink = nullptr;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(buf, 0, BYTES));
    CHECK(hipMemset(sink, 0, 64));
    CHECK(hipDeviceSynchronize());
    const dim3 block(256);
    hipLaunchKernelGGL(stream16, dim3(8192), block, 0, 0, (const uint4*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(stream8, dim3(8192), block, 0, 0, (const uint2*)buf, BYTES / 8, sink);
    hipLaunchKernelGGL(lane_records<8>, dim3(N_REC / 256), block, 0, 0, (const uint4*)buf, N_REC, sink);
    hipLaunchKernelGGL(lane_records<5>, dim3(N_REC / 256), block, 0, 0, (const uint4*)buf, N_REC, sink);
    hipLaunchKernelGGL(wave_records, dim3(N_REC / 4), block, 0, 0, (const uint4*)buf, N_REC, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    std::printf("{\"stream16\": %zu, \"stream8\": %zu, \"lane_records<8u>\": %zu, \"lane_records<5u>\": %zu, \"wave_records\": %zu}\n",
                BYTES, BYTES, (size_t)N_REC * 128, (size_t)N_REC * 80, (size_t)N_REC * 128);
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
