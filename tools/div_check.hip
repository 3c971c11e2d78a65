// Compares div_by_norm (raytracer_amd/csrc/dev_rt.h: three divisions by one magnitude through one shared reciprocal)
// with the compiler's own f64 divisions, bit for bit, over vectors whose components' exponents sweep a chosen range.
//   make raytracer_amd/div_check.bin     (hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off, as the library)
//   raytracer_amd/div_check.bin          (on the GPU box; status 1 if any quotient differs)
#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>
#include "../raytracer_amd/csrc/dev_rt.h"

__device__ uint64_t rng_next(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
// a double with a random sign and mantissa and an exponent drawn uniformly from [-span, span]; now and then an exact zero
__device__ double draw(uint64_t& s, int span, bool zeros) {
    const uint64_t m = rng_next(s);
    if (zeros && (m & 0xFF) == 0) return (m & 0x100) ? 0.0 : -0.0;
    const int e = (int)(rng_next(s) % (uint64_t)(2 * span + 1)) - span;
    const uint64_t bits = (m & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 + e) << 52);
    return __longlong_as_double((long long)bits);
}
__global__ void k_check(uint64_t seed, int span, int zeros, uint32_t per_thread, unsigned long long* out) {
    uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1) * 0x9E3779B97F4A7C15ull);
    unsigned long long bad = 0, fast = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        const D3 a = d3(draw(s, span, zeros), draw(s, span, zeros), draw(s, span, zeros));
        const double m = magnitude(a);
        const D3 q = div_by_norm(a, m);
        const D3 w = a / m;
        if (__double_as_longlong(q.x) != __double_as_longlong(w.x) || __double_as_longlong(q.y) != __double_as_longlong(w.y) ||
            __double_as_longlong(q.z) != __double_as_longlong(w.z)) {
            // NaN quotients (0/0) compare by bits too: both paths run the same plain division there
            bad++;
        }
        const double LO = 0x1p-200, HI = 0x1p200;
        if (m >= LO && m <= HI && fabs(a.x) >= LO && fabs(a.y) >= LO && fabs(a.z) >= LO) fast++;
    }
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], fast);
}
int main() {
    unsigned long long* d; unsigned long long h[2];
    if (hipMalloc((void**)&d, 16) != hipSuccess) { std::printf("no device memory\n"); return 2; }
    int rc = 0;
    const int spans[] = {2, 12, 60, 199, 210, 400, 1000};
    for (int zeros = 0; zeros < 2; zeros++)
        for (int span : spans) {
            (void)hipMemset(d, 0, 16);
            const uint32_t blocks = 4096, threads = 256, per = 256;
            hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + (uint64_t)span * 977 + (uint64_t)zeros, span, zeros, per, d);
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("launch failed\n"); return 2; }
            const double n = (double)blocks * threads * per;
            std::printf("exponents within +-%4d%s: %.0f vectors, %.1f %% through the shared reciprocal, %llu differ\n", span, zeros ? ", zeros mixed in" : "", n, 100.0 * (double)h[1] / n, h[0]);
            if (h[0]) rc = 1;
        }
    (void)hipFree(d);
    return rc;
}
