// Reading a count back between two dependent kernels: (a) hipMemcpyAsync of 8 bytes + hipStreamSynchronize, the way the stages do it, against (b) a one-wave kernel that
// posts the value and a sequence number into mapped pinned memory while the host spins on the sequence number.  Each round: kernel A (writes the count), the read-back,
// kernel B (launched by the host once it has the count).  Reported: time per round.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_latency fetch_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_a(uint64_t *cnt, uint64_t v) { if (threadIdx.x == 0) *cnt = v; }
__global__ void k_b(uint64_t *out, uint64_t n) { if (threadIdx.x == 0) *out = n; }
__global__ void k_post(const uint64_t *src, int n, volatile uint64_t *slot, uint64_t seq) {
    if ((int)threadIdx.x < n) slot[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) slot[8] = seq;
}
using clk = std::chrono::steady_clock;
int main() {
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint64_t *d, *o; (void)hipMalloc(&d, 64); (void)hipMalloc(&o, 64);
    uint64_t *slot; (void)hipHostMalloc(&slot, 128, hipHostMallocDefault);
    for (int i = 0; i < 16; i++) slot[i] = 0;
    const int N = 2000;
    for (int mode = 0; mode < 2; mode++) {
        uint64_t seq = 0, bad = 0;
        for (int rep = 0; rep < 2; rep++) {
            const auto t0 = clk::now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, d, (uint64_t)i + 7);
                uint64_t h = 0;
                if (mode == 0) { (void)hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); }
                else {
                    seq++;
                    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, (const uint64_t *)d, 1, (volatile uint64_t *)slot, seq);
                    while (__atomic_load_n(&slot[8], __ATOMIC_ACQUIRE) != seq) {}
                    h = slot[0];
                }
                if (h != (uint64_t)i + 7) bad++;
                hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, o, h);
            }
            (void)hipStreamSynchronize(st);
            if (rep) printf("%-44s %.2f us per round (kernel, read-back, kernel)%s\n", mode == 0 ? "hipMemcpyAsync(8 B) + hipStreamSynchronize" : "k_post into mapped memory + host spin",
                            std::chrono::duration<double, std::micro>(clk::now() - t0).count() / N, bad ? "  WRONG VALUES" : "");
        }
    }
    return 0;
}
