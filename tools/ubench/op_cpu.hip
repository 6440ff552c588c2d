// What does each kind of HIP call cost a rank in CPU time -- on the calling thread AND on the runtime's own threads?  (r5: a step of ~22 ms cost 15 ms of CPU on a thread of
// the runtime; which calls feed it?)  For each operation: N calls on one stream, one wait at the end; process CPU split into "this thread" and "all other threads".
// Build: hipcc --offload-arch=gfx950 -O3 -o op_cpu op_cpu.hip ; run: ./op_cpu [auto|spin|yield|blocking]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <vector>
__global__ void k_nop(int *p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_busy(unsigned long long ticks) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {} }
static double cpu_ms(clockid_t c) { timespec ts; clock_gettime(c, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
using clk = std::chrono::steady_clock;
int main(int argc, char **argv) {
    if (argc > 1) {
        const char *m = argv[1];
        const unsigned f = !strcmp(m, "spin") ? hipDeviceScheduleSpin : !strcmp(m, "yield") ? hipDeviceScheduleYield : !strcmp(m, "blocking") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        printf("--- hipSetDeviceFlags(%s) -> %s\n", m, hipGetErrorString(hipSetDeviceFlags(f)));
    }
    hipStream_t st, st2; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
    int *d = nullptr; (void)hipMalloc(&d, 1 << 20);
    void *pin = nullptr; (void)hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault);
    std::vector<char> pageable(1 << 20, 1);
    hipEvent_t ev, evt; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming); (void)hipEventCreate(&evt);
    std::vector<hipEvent_t> evs(4096);
    for (auto &e : evs) (void)hipEventCreate(&e);
    struct Op { const char *name; std::function<void(int)> f; };
    const Op ops[] = {
        {"kernel launch (empty)", [&](int) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr); }},
        {"kernel launch (20 us busy)", [&](int) { hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, st, 2000ull); }},
        {"hipMemsetAsync 4 KB", [&](int) { (void)hipMemsetAsync(d, 0, 4096, st); }},
        {"hipMemcpyAsync H2D 64 B pageable", [&](int) { (void)hipMemcpyAsync(d, pageable.data(), 64, hipMemcpyHostToDevice, st); }},
        {"hipMemcpyAsync H2D 64 KB pageable", [&](int) { (void)hipMemcpyAsync(d, pageable.data(), 65536, hipMemcpyHostToDevice, st); }},
        {"hipMemcpyAsync H2D 64 B pinned", [&](int) { (void)hipMemcpyAsync(d, pin, 64, hipMemcpyHostToDevice, st); }},
        {"hipMemcpyAsync D2H 64 B pinned", [&](int) { (void)hipMemcpyAsync(pin, d, 64, hipMemcpyDeviceToHost, st); }},
        {"hipMemcpyAsync D2H 64 B pageable", [&](int) { (void)hipMemcpyAsync(pageable.data(), d, 64, hipMemcpyDeviceToHost, st); }},
        {"hipMemcpyAsync D2D 64 KB", [&](int) { (void)hipMemcpyAsync(d, d + 65536, 65536, hipMemcpyDeviceToDevice, st); }},
        {"hipEventRecord (timing event, fresh)", [&](int i) { (void)hipEventRecord(evs[(size_t)i & 4095], st); }},
        {"hipEventRecord (no-timing, same)", [&](int) { (void)hipEventRecord(ev, st); }},
        {"record + hipStreamWaitEvent on stream 2", [&](int) { (void)hipEventRecord(ev, st); (void)hipStreamWaitEvent(st2, ev, 0); }},
        {"kernel + hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, st, 2000ull); (void)hipStreamSynchronize(st); }},
        {"kernel (1 ms) + hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, st, 100000ull); (void)hipStreamSynchronize(st); }},
    };
    {   // a cross-stream dependency that stays pending while a long kernel runs: who pays for it?  (the calling thread sleeps; nobody spins on purpose)
        const int N = 40;
        for (int mode = 0; mode < 3; mode++) {
            (void)hipDeviceSynchronize();
            const double p0 = cpu_ms(CLOCK_PROCESS_CPUTIME_ID), h0 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
            const auto t0 = clk::now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, st, 500000ull);      // 5 ms
                if (mode == 1) { (void)hipEventRecord(ev, st); (void)hipStreamWaitEvent(st2, ev, 0); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st2, (int *)nullptr); }
                if (mode == 2) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int *)nullptr); }
                timespec ts = {0, 6000000}; nanosleep(&ts, nullptr);
            }
            const double h1 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
            (void)hipDeviceSynchronize();
            const double wall = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            const double p1 = cpu_ms(CLOCK_PROCESS_CPUTIME_ID), h2 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
            printf("%-60s %8.2f us wall/op | calling thread %7.2f us/op | other threads %8.2f us/op\n",
                   mode == 0 ? "5 ms kernel; the caller sleeps 6 ms" : mode == 1 ? "5 ms kernel, stream 2 waits for it (event) + kernel; sleep" : "5 ms kernel + dependent kernel on the SAME stream; sleep",
                   wall / N, (h1 - h0) * 1e3 / N, ((p1 - p0) - (h2 - h0)) * 1e3 / N);
        }
    }
    for (const Op &op : ops) {
        const bool slow = strstr(op.name, "1 ms") != nullptr;
        const int N = slow ? 200 : 2000;
        for (int i = 0; i < 50; i++) op.f(i);
        (void)hipDeviceSynchronize();
        const auto t0 = clk::now();
        const double p0 = cpu_ms(CLOCK_PROCESS_CPUTIME_ID), h0 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
        for (int i = 0; i < N; i++) op.f(i);
        const double h1 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
        (void)hipDeviceSynchronize();
        const double wall = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
        const double p1 = cpu_ms(CLOCK_PROCESS_CPUTIME_ID), h2 = cpu_ms(CLOCK_THREAD_CPUTIME_ID);
        printf("%-44s %8.2f us wall/op | calling thread %7.2f us/op (+%.0f us in the final wait) | other threads %7.2f us/op\n", op.name, wall / N, (h1 - h0) * 1e3 / N, (h2 - h1) * 1e3,
               ((p1 - p0) - (h2 - h0)) * 1e3 / N);
    }
    return 0;
}
