// Round trip of "launch a tiny kernel, wait for it": hipStreamSynchronize vs spinning on hipEventQuery vs hipEventSynchronize.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop(int *p) { if (p && threadIdx.x == 999) *p = 1; }
using clk = std::chrono::steady_clock;
int main() {
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int N = 2000;
    for (int mode = 0; mode < 3; mode++) {
        for (int w = 0; w < 100; w++) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr); (void)hipStreamSynchronize(st); }
        auto t0 = clk::now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr);
            if (mode == 0) (void)hipStreamSynchronize(st);
            else if (mode == 1) { (void)hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} }
            else { (void)hipEventRecord(ev, st); (void)hipEventSynchronize(ev); }
        }
        const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / N;
        printf("%-28s %.2f us per launch + wait\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventQuery spin" : "hipEventSynchronize", us);
    }
    // a copy of 4 bytes device -> host and the wait (what the count read-backs between stages are)
    int *d; (void)hipMalloc(&d, 4); int h = 0;
    auto t0 = clk::now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d); (void)hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); }
    printf("%-28s %.2f us per launch + 4-byte read-back + wait\n", "kernel, D2H, sync", std::chrono::duration<double, std::micro>(clk::now() - t0).count() / N);
    return 0;
}
