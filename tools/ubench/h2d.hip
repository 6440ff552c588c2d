// Host -> device staging micro-benchmark: what a pinned 8 MiB block costs to fill (memcpy from pageable memory) and to send
// (hipMemcpyAsync), per hipHostMalloc flavour.  Built as a shared object so that a Python driver can load it before or after
// torch (which brings its own HIP runtime): tools/ubench/h2d_driver.py.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

__global__ void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

extern "C" int h2d_run(void) {
    const size_t CH = 8u << 20, TOTAL = 512u << 20;
    int rtv = 0;
    (void)hipRuntimeGetVersion(&rtv);
    printf("hip runtime version %d\n", rtv);
    char *src = (char *)malloc(TOTAL);
    memset(src, 'A', TOTAL);
    void *dev = nullptr;
    if (hipMalloc(&dev, TOTAL) != hipSuccess) return 1;
    struct Fl { const char *name; unsigned f; };
    const Fl fl[] = {{"default", hipHostMallocDefault}, {"noncoherent", hipHostMallocNonCoherent}, {"coherent", hipHostMallocCoherent},
                     {"portable", hipHostMallocPortable}, {"numa_user", hipHostMallocNumaUser}, {"writecombined", hipHostMallocWriteCombined}};
    for (const Fl &f : fl) {
        void *blk[8] = {};
        bool ok = true;
        for (auto &b : blk) if (hipHostMalloc(&b, CH, f.f) != hipSuccess) { ok = false; (void)hipGetLastError(); break; }
        if (!ok) { printf("%-14s hipHostMalloc failed\n", f.name); continue; }
        for (auto b : blk) memset(b, 0, CH);
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        // fill only
        auto t0 = clk::now();
        for (size_t o = 0, k = 0; o < TOTAL; o += CH, k ^= 1) memcpy(blk[k], src + o, CH);
        const double fill = since(t0);
        // send only
        t0 = clk::now();
        for (size_t o = 0, k = 0; o < TOTAL; o += CH, k ^= 1) (void)hipMemcpyAsync((char *)dev + o, blk[k], CH, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        const double send = since(t0);
        // the library's scheme: T threads, two blocks and one stream each
        double piped[4] = {0, 0, 0, 0};
        int ti = 0;
        for (int mode = 0; mode < 2; mode++)
        for (int T : {1, 4}) {
            t0 = clk::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t]() {
                    hipStream_t s2; hipEvent_t ev[2];
                    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
                    (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
                    bool used[2] = {false, false};
                    int k = 0;
                    for (size_t o = (size_t)t * CH; o < TOTAL; o += (size_t)T * CH, k ^= 1) {
                        if (used[k]) (void)hipEventSynchronize(ev[k]);
                        memcpy(blk[t * 2 + k], src + o, CH);
                        if (mode == 0) (void)hipMemcpyAsync((char *)dev + o, blk[t * 2 + k], CH, hipMemcpyHostToDevice, s2);
                        else hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s2, (uint4 *)((char *)dev + o), (const uint4 *)blk[t * 2 + k], CH / 16);
                        (void)hipEventRecord(ev[k], s2);
                        used[k] = true;
                    }
                    (void)hipStreamSynchronize(s2);
                    (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]); (void)hipStreamDestroy(s2);
                });
            for (auto &x : th) x.join();
            piped[ti++] = since(t0);
        }
        printf("%-14s fill %6.1f GB/s   send %6.1f GB/s   staged, hipMemcpyAsync: 1 thread %6.1f GB/s, 4 threads %6.1f GB/s   staged, copy kernel: 1 thread %6.1f GB/s, 4 threads %6.1f GB/s\n",
               f.name, TOTAL / fill / 1e9, TOTAL / send / 1e9, TOTAL / piped[0] / 1e9, TOTAL / piped[1] / 1e9, TOTAL / piped[2] / 1e9, TOTAL / piped[3] / 1e9);
        (void)hipStreamDestroy(st);
        for (auto b : blk) (void)hipHostFree(b);
    }
    // pageable straight to the device, and a registered source
    auto t0 = clk::now();
    (void)hipMemcpy(dev, src, TOTAL, hipMemcpyHostToDevice);
    printf("pageable hipMemcpy %6.1f GB/s\n", TOTAL / since(t0) / 1e9);
    t0 = clk::now();
    if (hipHostRegister(src, TOTAL, hipHostRegisterDefault) == hipSuccess) {
        const double reg = since(t0);
        t0 = clk::now();
        (void)hipMemcpy(dev, src, TOTAL, hipMemcpyHostToDevice);
        const double cp = since(t0);
        t0 = clk::now();
        (void)hipHostUnregister(src);
        printf("hipHostRegister %6.1f ms, copy %6.1f GB/s, unregister %6.1f ms  (register+copy+unregister: %6.1f GB/s)\n", reg * 1e3, TOTAL / cp / 1e9, since(t0) * 1e3,
               TOTAL / (reg + cp + since(t0)) / 1e9);
    } else { (void)hipGetLastError(); printf("hipHostRegister failed\n"); }
    (void)hipFree(dev);
    free(src);
    return 0;
}
