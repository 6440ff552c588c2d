// Measurement aid (r6): host-to-device rate of a 362 MB block of PINNED host memory -- the runtime's copy (one call, 4 MB pieces) against a kernel that reads the
// host pages itself (16-byte loads from the mapped pointer).  build: hipcc --offload-arch=gfx950 -O3 -o h2d_kernel_copy h2d_kernel_copy.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main() {
    const size_t N = 362u << 20;
    void *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(&h, N, hipHostMallocDefault));
    memset(h, 7, N);
    CK(hipMalloc(&d, N));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void *hd = nullptr;
    CK(hipHostGetDevicePointer(&hd, h, 0));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = now();
        CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
        auto t1 = now();
        for (size_t o = 0; o < N; o += 4u << 20) CK(hipMemcpyAsync((char *)d + o, (char *)h + o, std::min<size_t>(4u << 20, N - o), hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        printf("runtime copy: one call %.2f ms = %.1f GB/s; 4 MB pieces %.2f ms = %.1f GB/s\n", ms(t0, t1), N / ms(t0, t1) / 1e6, ms(t1, t2), N / ms(t1, t2) / 1e6);
    }
    for (int grid : {16, 32, 64, 128, 256, 512, 1024}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = now();
            hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (uint4 *)d, (const uint4 *)hd, N / 16);
            CK(hipStreamSynchronize(st));
            best = std::min(best, ms(t0, now()));
        }
        printf("kernel copy, %4d workgroups: %.2f ms = %.1f GB/s\n", grid, best, N / best / 1e6);
    }
    // the same in 4 MB pieces (a launch per piece)
    {
        auto t0 = now();
        for (size_t o = 0; o < N; o += 4u << 20) hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, st, (uint4 *)((char *)d + o), (const uint4 *)((char *)hd + o), std::min<size_t>(4u << 20, N - o) / 16);
        CK(hipStreamSynchronize(st));
        printf("kernel copy, 4 MB pieces of 64 workgroups: %.2f ms = %.1f GB/s\n", ms(t0, now()), N / ms(t0, now()) / 1e6);
    }
    return 0;
}
