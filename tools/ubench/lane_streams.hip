// micro-benchmark: every lane of a wave appends 16 B per step to a stream of its own (the store pattern of a DP kernel that keeps one read per lane:
// 64 streams per wave, 16 B per stream and step), with `valu` dependent VALU instructions between two stores.  Question: what write rate do such
// per-lane streams reach on MI355X, and how many waves does it take?
// Build: hipcc --offload-arch=gfx950 -O3 -o lane_streams lane_streams.hip ; run: ./lane_streams
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VALU, int GROUP>
__global__ void __launch_bounds__(64) k(uint4 *out, int64_t steps) {
    const int64_t stream = (int64_t)blockIdx.x * 64 + threadIdx.x;
    uint4 *p = out + stream * steps;
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 5;
    for (int64_t s = 0; s < steps; s += GROUP) {
        uint4 v[GROUP];
#pragma unroll
        for (int g = 0; g < GROUP; g++) {
#pragma unroll
            for (int u = 0; u < VALU / 4; u++) {
                asm volatile("v_xor_b32 %0, %0, %1\n\tv_and_b32 %1, %1, %2\n\tv_or_b32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
            v[g] = make_uint4(a, b, c, d);
        }
#pragma unroll
        for (int g = 0; g < GROUP; g++) p[s + g] = v[g];
    }
}

template <int VALU, int GROUP>
static void run(uint4 *d, int waves, int64_t steps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<VALU, GROUP>), dim3(waves), dim3(64), 0, 0, d, steps / 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<VALU, GROUP>), dim3(waves), dim3(64), 0, 0, d, steps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double bytes = (double)waves * 64 * steps * 16;
    printf("waves %5d  steps %6lld  valu/step %3d  stores grouped by %d  %8.3f ms  %7.1f GB/s  (%.2f GB)  %6.1f cycles/step @2.4 GHz\n", waves, (long long)steps, VALU, GROUP, best, bytes / best * 1e-6, bytes * 1e-9,
           best * 1e-3 * 2.4e9 / steps);
}

int main() {
    CK(hipSetDevice(0));
    const int64_t steps = 30720;
    uint4 *d;
    CK(hipMalloc(&d, (size_t)2560 * 64 * steps * 16));
    for (int waves : {625, 1250, 2500}) {
        run<0, 1>(d, waves, steps);
        run<0, 4>(d, waves, steps);
        run<48, 1>(d, waves, steps);
        run<100, 1>(d, waves, steps);
        run<100, 4>(d, waves, steps);
        run<100, 8>(d, waves, steps);
    }
    CK(hipFree(d));
    return 0;
}
